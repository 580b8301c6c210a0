"""bench.py's output contract, checked on the lines committed under profiles/ (the bench itself needs a GPU):
every line carries the driver's keys plus `roofline` and `cpu_baseline`, and the numbers are self-consistent."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _lines():
    out = []
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_bench_v7*.json")) +
                    glob.glob(os.path.join(ROOT, "profiles", "r01_bench_minkunet34c_*.json")) +
                    glob.glob(os.path.join(ROOT, "profiles", "r02_bench_final*.json"))):
        with open(p) as f:
            out.append((os.path.basename(p), json.loads(f.read().strip().splitlines()[-1])))
    return out


def test_committed_bench_lines_follow_the_contract():
    lines = _lines()
    assert len(lines) >= 5
    for name, d in lines:
        for k in REQUIRED:
            assert k in d, (name, k)
        assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
        assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (name, k)
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        # the roof follows the launch's arithmetic intensity (round 2): matrix peak of the dtype, or the 8 TB/s of HBM
        assert d["dtype"] in ("f32", "bf16")
        assert r["peak"] == ((2500.0 if d["dtype"] == "bf16" else 157.3) if r["bound"] == "mfma" else 8000.0)
        assert r["unit"] == ("TFLOP/s" if r["bound"] == "mfma" else "GB/s")
        # value = voxels of all ranks / step time
        pts = d["config"]["points_per_gpu"] * d["n_gpus"]
        assert abs(d["value"] - pts / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
        if d["cpu_baseline"] is not None:
            c = d["cpu_baseline"]
            assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_round_2_headline_line():
    """The closing line of round 2: config 2 in fp32 on one GPU, the reference's own CPU operators timed beside it,
    the forward kernel priced against the fp32 MFMA peak (and, since it issues to the bf16 pipe, against that pipe's
    ceiling for six MFMAs per product), HBM traffic from the committed PMC pass of that kernel."""
    d = dict(_lines())["r02_bench_final.json"]
    assert d["dtype"] == "f32" and d["n_gpus"] == 1
    assert "100000 voxels" in d["config"]["workload"] and "64->128" in d["config"]["workload"]
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["cores"] >= 1
    r = d["roofline"]
    assert r["kernel"].startswith("k_conv_tile_f32x3") and r["bound"] == "mfma" and r["peak"] == 157.3
    assert abs(r["peak_of_pipe"] - 2500.0 / 6) < 0.1 and abs(r["frac_of_pipe"] - r["achieved"] / r["peak_of_pipe"]) < 1e-3
    assert r["traffic"] > r["compulsory_bytes_per_launch"]
    assert len(d["timing"]["blocks_ms_per_step"]) >= 3 and min(d["timing"]["blocks_ms_per_step"]) == d["ms_per_step"]
    two = dict(_lines())["r02_bench_final_n2_selfspawn_1gpu.json"]
    assert two["n_gpus"] == 2 and two["config"]["oversubscribed"] is True      # two ranks on the one GPU of the box


def test_headline_line_is_config_2_in_fp32():
    d = dict(_lines())["r01_bench_v7.json"]
    assert d["dtype"] == "f32" and d["n_gpus"] == 1
    assert "100000 voxels" in d["config"]["workload"] and "64->128" in d["config"]["workload"]
    assert d["cpu_baseline"]["kind"] == "reference"
    assert d["roofline"]["traffic"] > 0 and d["roofline"]["kernel"].startswith("k_conv_tile_f32")


def test_multi_gpu_line_carries_the_exchange_and_the_ddp_workload():
    """under --gpus N > 1 the line prices the gradient exchange (per-rank times, the step without the
    all-reduce, the all-reduce on its own, what the overlapped step still pays, an accumulation window), names the
    collective layer as torch.distributed reports it, and attaches BASELINE configs[3] — MinkUNet34C under DDP — as
    `workloads.minkunet34c_bf16_ddp`.  Checked on the committed 2-ranks-on-one-GPU run (gloo; RCCL refuses two ranks on
    one device): profiles/r04_bench_n2_1gpu_gloo.json."""
    with open(os.path.join(ROOT, "profiles", "r04_bench_n2_1gpu_gloo.json")) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["oversubscribed"] is True and d["scaling"] == "weak"
    for entry, grad_bytes in ((d, 27 * 64 * 128 * 4), (d["workloads"]["minkunet34c_bf16_ddp"], 37856052 * 4)):
        m = entry["multi_gpu"]
        assert m["backend"] == "gloo" and m["world_size_reported"] == 2
        assert len(m["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in m["per_rank_ms_per_step"])
        assert m["allreduce_ms"]["bytes"] == grad_bytes and m["allreduce_ms"]["standalone"] > 0
        assert 0 < m["no_sync_ms_per_step"] <= entry["ms_per_step"] * 1.05
        assert abs(m["parallel_efficiency"] - m["no_sync_ms_per_step"] / entry["ms_per_step"]) < 0.01
        assert abs(m["allreduce_ms"]["exposed_in_step"] - (entry["ms_per_step"] - m["no_sync_ms_per_step"])) < 0.02 * entry["ms_per_step"]
    acc = d["multi_gpu"]["accumulation"]
    assert acc["window"] == 4 and acc["ms_per_micro_step"] < d["ms_per_step"]      # 3 of 4 micro-steps skip the exchange
    w = d["workloads"]["minkunet34c_bf16_ddp"]
    assert w["n_gpus"] == 2 and w["dtype"] == "bf16" and "DDP" in w["config"]["parallelism"]
    assert w["multi_gpu"]["per_rank_points"] == [200000, 200000]
    pts = 2 * 200000
    assert abs(w["value"] - pts / (w["ms_per_step"] * 1e-3) / 1e6) < 0.01 * w["value"]
