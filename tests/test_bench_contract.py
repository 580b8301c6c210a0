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
                    glob.glob(os.path.join(ROOT, "profiles", "r01_bench_minkunet34c_*.json"))):
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
        assert d["dtype"] in ("f32", "bf16") and r["peak"] == (2500.0 if d["dtype"] == "bf16" else 157.3)
        # value = voxels of all ranks / step time
        pts = d["config"]["points_per_gpu"] * d["n_gpus"]
        assert abs(d["value"] - pts / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
        if d["cpu_baseline"] is not None:
            c = d["cpu_baseline"]
            assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_headline_line_is_config_2_in_fp32():
    d = dict(_lines())["r01_bench_v7.json"]
    assert d["dtype"] == "f32" and d["n_gpus"] == 1
    assert "100000 voxels" in d["config"]["workload"] and "64->128" in d["config"]["workload"]
    assert d["cpu_baseline"]["kind"] == "reference"
    assert d["roofline"]["traffic"] > 0 and d["roofline"]["kernel"].startswith("k_conv_tile_f32")
