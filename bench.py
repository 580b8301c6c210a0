#!/usr/bin/env python
"""bench.py — the driver's benchmark contract for the MI355X sparse-convolution hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic input: forward + backward of ONE
MinkowskiConvolution (3-D, k = 3, s = 1, 64 -> 128 channels, fp32, no bias) on a 100k-voxel scene per
GPU (BASELINE.json configs[1]), kernel map cached (steady-state training layer), inputs resident in
HBM.  With N > 1 every rank owns its own scene (weak scaling, no data-path collective) and the gradients
are averaged by torch DistributedDataParallel over RCCL (bucketed, overlapped with the backward pass).
value = total voxels of all ranks / max-over-ranks step time.

Launch: under torch.distributed.run the ranks come from the environment (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_*).  A plain `python bench.py --gpus N` with N > 1 starts the N ranks itself (one process per GPU,
127.0.0.1 rendezvous).  When the box has fewer GPUs than ranks (a 1-GPU test box), the ranks share the
GPUs and talk over gloo instead of RCCL; the line then says "oversubscribed": true and is a functional
check of the N > 1 path, not a scaling number.

Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both
sides and reduced with max over ranks; blocks are repeated until at least --min-time seconds have been
timed (a 20-step block of the headline workload is 7 ms) and the MEDIAN block is reported (`ms_per_step`, `value`;
round 2 reported the fastest block, which is still listed as `fastest_block_ms_per_step`); every block is under
`blocks_ms_per_step`.

Other workloads (never the default; parity-test configurations timed for DESIGN.md):
    --workload conv4d     BASELINE configs[4]: 4-D k = 3 (K = 81), 400k voxels in 100^3 x 8 frames, 32 -> 64
    --workload minkunet   BASELINE configs[2]/[3]: MinkUNet34C forward + loss + backward + SGD step on the
                          200k-voxel plane-union scene of SURVEY 8(d); one scene per rank, DDP

Rank 0 prints ONE JSON line carrying `roofline` (dominant kernel, HIP-event timed inside the timed region
on the launch stream) and `cpu_baseline` (the reference's own CPU operators, oracle/_ref, timed on this
box's host cores at N = 1 on a bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0
RIDGE_F32 = PEAK_F32_MATRIX_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)    # ~19.7 flop/B
RIDGE_BF16 = PEAK_BF16_MATRIX_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)  # ~312 flop/B


# ------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torch.distributed.run
# ------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """Start one process per rank of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, 127.0.0.1 rendezvous) and wait for them; rank 0 prints the JSON line on our stdout."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


import torch  # noqa: E402


def pmc_traffic(kernel, n, extent, cin, cout):
    """HBM-side bytes per launch of `kernel`: a CITED constant from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json, collected with scripts/archive/gpu_pmc_traffic.sh on this exact workload in their own
    runs — PMC passes cannot share a run with the timed region), or None for any other workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        w = t["workload"]
        if (w["points"], w["extent"], w["cin"], w["cout"]) != (n, extent, cin, cout):
            return None, None
        k = t["kernels"][kernel]
        return int(k["fetch_bytes"] + k["write_bytes"]), k.get("source", t.get("source", "profiles/pmc_traffic.json"))
    except Exception:  # noqa: BLE001
        return None, None


# ------------------------------------------------------------------------------------------------------
# HBM-side traffic, measured in THIS run (round 4): rocprofv3 --pmc passes over a child process
# ------------------------------------------------------------------------------------------------------
PMC_CONV_KERNELS = ("k_conv_tile", "k_wgrad", "k_conv_splitk_reduce", "k_conv_off", "k_conv_halo", "k_conv_stem")


def pmc_pass(workload, dtype, counters, steps, timeout_s):
    """One rocprofv3 counter pass over a CHILD process of this script that runs `steps` steps of `workload` (counters
    cannot share a run with the timed region: a PMC pass serialises the kernels).  -> {kernel name: {counter: (sum over
    launches, launches)}} or None.  FETCH_SIZE / WRITE_SIZE need separate passes (MI355X_MICROARCH.md: TCC has 4
    slots, FETCH_SIZE costs 3, WRITE_SIZE 2)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="me_pmc_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--workload", workload, "--dtype", dtype, "--pmc-child", str(steps)]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
        agg = {}
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = agg.setdefault(row.get("Kernel_Name", ""), {})
                    c = k.setdefault(row["Counter_Name"], [0.0, 0])
                    c[0] += float(row["Counter_Value"])
                    c[1] += 1
        # durations of the same (serialised, profiled) launches: the counter passes' own clock — never compared with the
        # timed region (MI355X_MICROARCH.md: profiled passes run at another clock)
        for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    try:
                        d = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    except (KeyError, ValueError):
                        continue
                    k = agg.setdefault(row.get("Kernel_Name", ""), {})
                    c = k.setdefault("_duration_ns", [0.0, 0])
                    c[0] += d
                    c[1] += 1
        return agg or None
    except Exception:  # noqa: BLE001  (a missing profiler / a timeout must not cost the benchmark line)
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measure_traffic(workload, dtype, kernel_match, steps, deadline, extra_steps=0):
    """HBM-side bytes of the launches whose kernel name contains every string of `kernel_match` (None: all convolution
    kernels): FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM) + WRITE_SIZE, both in
    KiB.  -> dict(per_launch bytes, per_step bytes, launches_per_step, fetch / write split) or None when the profiler is
    absent, a pass failed or the time budget (`deadline`, perf_counter seconds) is used up."""
    res, per_kernel = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = deadline - time.perf_counter()
        if left < 20:
            return None
        agg = pmc_pass(workload, dtype, [counter], steps, min(left, 150))
        if not agg:
            return None
        tot, n = 0.0, 0
        for name, cs in agg.items():
            if counter not in cs:
                continue
            if any(m in name for m in PMC_CONV_KERNELS):      # every convolution kernel of the pass, by name
                e = per_kernel.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
                e[counter] += cs[counter][0] * 1024.0
                e["launches"] = max(e["launches"], cs[counter][1])
            hit = all(m in name for m in kernel_match) if kernel_match else any(m in name for m in PMC_CONV_KERNELS)
            if hit:
                tot += cs[counter][0]
                n += cs[counter][1]
        if n == 0:
            return None
        res[counter] = (tot * 1024.0, n)
    fetch = 2.0 * res["FETCH_SIZE"][0]
    write = res["WRITE_SIZE"][0]
    n = res["FETCH_SIZE"][1]
    ran = steps + extra_steps     # (the convolution workloads run one cold forward + backward before their steps)
    by_kernel = {}
    for name, e in per_kernel.items():
        short = name.split("(")[0].replace("void ", "").replace("me::", "")
        by_kernel[short] = {"launches_per_step": round(e["launches"] / ran, 2),
                            "bytes_per_launch": int((2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) / max(e["launches"], 1)),
                            "fetch_bytes_per_launch": int(2.0 * e["FETCH_SIZE"] / max(e["launches"], 1)),
                            "write_bytes_per_launch": int(e["WRITE_SIZE"] / max(e["launches"], 1))}
    return {"per_launch": int((fetch + write) / n), "per_step": int((fetch + write) / ran),
            "launches_per_step": round(n / ran, 2), "fetch_bytes_per_launch": int(fetch / n),
            "write_bytes_per_launch": int(write / n), "by_kernel": by_kernel,
            "source": f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each over a child process "
                      f"running {steps} steps of the workload (FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 B)"}


SQ_COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
               "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU_MFMA_MOPS_BF16"]
N_SIMDS, N_XCDS = 1024, 8


def measure_sq(workload, dtype, steps, deadline):
    """Counter-based matrix-pipe utilisation of every convolution kernel of the workload (north_star
    "rocprof MFMA utilisation reported"): ONE rocprofv3 pass (7 SQ counters + GRBM_GUI_ACTIVE: within the 8 SQ slots of
    MI355X_MICROARCH.md) over a child process.  Per kernel, averages per launch:
      mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) — rocprofiler's own MfmaUtil
                       (counter_defs.yaml: reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum) / (reduce(GRBM_GUI_ACTIVE,max) * SIMD_NUM));
      wave_wait_frac / issue_stall_frac / active_frac = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over
                       SQ_WAVE_CYCLES (disjoint shares of a wave's life: parked in s_waitcnt / barrier, stalled at
                       issue, issuing).
    -> {short kernel name: {...}} or None."""
    left = deadline - time.perf_counter()
    if left < 25:
        return None
    agg = pmc_pass(workload, dtype, SQ_COUNTERS, steps, min(left, 150))
    if not agg:      # (a counter this profiler build does not know: the three that define MfmaUtil alone)
        left = deadline - time.perf_counter()
        if left < 25:
            return None
        agg = pmc_pass(workload, dtype, SQ_COUNTERS[:3], steps, min(left, 150))
    if not agg:
        return None
    out = {}
    for name, cs in agg.items():
        if not any(m in name for m in PMC_CONV_KERNELS) or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs:
            continue
        n = max(cs["SQ_VALU_MFMA_BUSY_CYCLES"][1], 1)
        per = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        cycles = per.get("GRBM_GUI_ACTIVE", 0.0) / N_XCDS
        short = name.split("(")[0].replace("void ", "").replace("me::", "")
        e = {"launches": n, "mfma_busy_frac": round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * N_SIMDS), 4) if cycles else None,
             "gpu_cycles": int(cycles)}
        wc = per.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            e["wave_wait_frac"] = round(per.get("SQ_WAIT_ANY", 0.0) / wc, 4)
            e["issue_stall_frac"] = round(per.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
            e["active_frac"] = round(per.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
        if "SQ_BUSY_CYCLES" in per and cycles:
            e["sq_busy_frac"] = round(per["SQ_BUSY_CYCLES"] / 32.0 / cycles, 4)      # (32 shader engines)
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in per:
            e["mfma_mops_bf16"] = int(per["SQ_INSTS_VALU_MFMA_MOPS_BF16"])
        if "_duration_ns" in per:
            e["profiled_us"] = round(per["_duration_ns"] / 1e3, 2)
            if cycles:
                e["profiled_clock_ghz"] = round(cycles / per["_duration_ns"], 3)
        out[short] = e
    return out or None


def attach_sq(line, args, deadline):
    """roofline.mfma_busy_frac (+ by_pass / by_kernel tables) from measure_sq, for the headline and — budget permitting —
    the MinkUNet34C entry (weighted by launches x cycles over its convolution kernels)"""
    if args.pmc == "off":
        return
    r = line.get("roofline") or {}
    tag = r.get("pmc_match")
    if tag:
        sq = measure_sq(args.workload, args.dtype, 3, deadline)
        if sq:
            rows = {}
            for short, e in sq.items():
                which = "forward" if all(m in short + "(" for m in tag) else \
                    ("wgrad_reduce" if "reduce" in short else ("wgrad" if "wgrad" in short else "dgrad"))
                rows[which if args.workload != "minkunet" else short] = dict(e, kernel=short)
            r["mfma_busy_frac"] = (rows.get("forward") or {}).get("mfma_busy_frac")
            r["mfma_busy_by_pass"] = rows
            r["mfma_busy_note"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ... in this run, one pass over a "
                                   "child process: busy cycles of the matrix pipes / (kernel cycles x 1024 SIMDs) = rocprofiler's "
                                   "MfmaUtil; `frac` beside it is flops / time / peak")
    ent = (line.get("workloads") or {}).get("minkunet34c_bf16_200k")
    if isinstance(ent, dict) and "roofline" in ent:
        sq = measure_sq("minkunet", "bf16", 2, deadline)
        if sq:
            tot_busy = sum(e["mfma_busy_frac"] * e["gpu_cycles"] * e["launches"] for e in sq.values() if e.get("mfma_busy_frac"))
            tot_cyc = sum(e["gpu_cycles"] * e["launches"] for e in sq.values() if e.get("mfma_busy_frac") is not None)
            ent["roofline"]["mfma_busy_frac"] = round(tot_busy / tot_cyc, 4) if tot_cyc else None
            top = sorted(sq.items(), key=lambda kv: -kv[1]["gpu_cycles"] * kv[1]["launches"])[:12]
            ent["roofline"]["mfma_busy_by_kernel"] = {k: v for k, v in top}
            ent["roofline"]["mfma_busy_note"] = "all convolution kernels of the step, cycle-weighted; the 12 heaviest listed"


def attach_traffic(line, args, deadline):
    """fill line['roofline']['traffic'] (and the compact workload entries) from in-run PMC passes, most important first,
    while the budget lasts; entries that are not reached keep their cited constant / null"""
    if args.pmc == "off":
        return
    r = line.get("roofline") or {}
    tag = r.get("pmc_match")
    if tag:
        t = measure_traffic(args.workload, args.dtype, tag, 3, deadline, extra_steps=0 if args.workload == "minkunet" else 1)
        if t:
            r["traffic"] = t["per_launch"]
            r["traffic_detail"] = t
            r["traffic_note"] = "HBM-side bytes per launch of this kernel — " + t["source"]
            r["traffic_over_compulsory"] = round(t["per_launch"] / r["compulsory_bytes_per_launch"], 3) \
                if r.get("compulsory_bytes_per_launch") else None
            # the same passes, every convolution kernel of the step: forward (this kernel), input gradient (the other tile
            # kernel), weight gradient (+ its reduce) against SURVEY 8(d)'s compulsory bytes of each
            comp = r.get("compulsory_by_pass")
            if comp and t.get("by_kernel"):
                rows = {}
                for short, e in t["by_kernel"].items():
                    which = "forward" if all(m in short + "(" for m in tag) else \
                        ("wgrad_reduce" if "reduce" in short else ("wgrad" if "wgrad" in short else "dgrad"))
                    rows[which] = dict(e, kernel=short,
                                       over_compulsory=(round(e["bytes_per_launch"] / comp[which], 3) if comp.get(which) else None))
                r["traffic_by_pass"] = rows
    for name, wl, dt in (("minkunet34c_bf16_200k", "minkunet", "bf16"), ("conv4d_f32_400k", "conv4d", "f32")):
        ent = (line.get("workloads") or {}).get(name)
        if not isinstance(ent, dict) or "roofline" not in ent:
            continue
        rr = ent["roofline"]
        t = measure_traffic(wl, dt, rr.get("pmc_match"), 2 if wl == "minkunet" else 3, deadline,
                            extra_steps=0 if wl == "minkunet" else 1)
        if t:
            rr["traffic"] = t["per_step"] if wl == "minkunet" else t["per_launch"]
            rr["traffic_detail"] = t
            rr["traffic_note"] = ("HBM-side bytes per STEP of all convolution launches — " if wl == "minkunet" else
                                  "HBM-side bytes per launch of this kernel — ") + t["source"]



def arena_per_rank(ex, dist_utils, dev):
    """what every rank's GradientArena did in its last step — bytes of its flat buffers, gradients born in place, copied
    in, pieces, pieces all-reduced from inside the backward pass — as lists over the ranks (the first
    SCALE run explains itself: a rank that copies gradients in, or whose pieces fall back to the closing all-reduce, shows
    up here)."""
    if ex.arena is None:
        return None
    d = ex.arena.describe()
    own = {"bytes": float(sum(d["buffers"].values())), "born_in_place": float(d["born_in_place"]),
           "copied_in": float(d["copied_in"]), "pieces": float(d["pieces"]),
           "overlapped_pieces": float(d["overlapped_pieces"])}
    return {k: [int(v) for v in dist_utils.gather_over_ranks(own[k], dev)] for k in own}



def hip_timed(fn):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    torch.cuda.synchronize()
    return r, s.elapsed_time(e)


def cold_path(ME, MEB, feats, coords, dev, n, D=3, K=27, cin=64, cout=128, bf16=False):
    """Coordinate insertion + kernel-map build + tile plans of one scene, each timed with HIP events (best of
    three: the first repetition pays the allocator), with the achieved rate on SURVEY 8(d)'s algorithmic bytes
    (probes N*K x key bytes + 8 B per pair written; insert N x key bytes + table + maps)."""
    out, best = {}, {}
    km = None
    native = ME.is_native()
    B = ME.host.backend()
    n_pairs = 0
    for rep in range(3):
        mgr = B.CoordinateMapManagerGPU_c10()
        (key, _), t_ins = hip_timed(lambda: mgr.insert_and_map(coords, [1] * D, ""))
        if native:
            # (the native build enqueues probe + compaction and copies the pair counts asynchronously; the counts are
            # read right here, as a first launch on the new map would)
            n_pairs, t_km = hip_timed(lambda: mgr.kernel_map_pairs(key, key, [3] * D, [1] * D, [1] * D, 0, False, False))
        else:
            km, t_km = hip_timed(lambda: mgr._kernel_map(key, key, [3] * D, [1] * D, [1] * D, ME.RegionType.HYPER_CUBE,
                                                         None, False, False))
            n_pairs = km.n_pairs

        def plans():
            # the launch configurations of forward and dgrad: tile plan (+ the spatial index behind its tile order)
            for tgt, (cs, cd) in (("out", (cin, cout)), ("in", (cout, cin))):
                if native:
                    mgr._conv_cfg(key, key, [3] * D, [1] * D, [1] * D, 0, False, tgt, cs, cd, bf16)
                else:
                    MEB._conv_launch_cfg(km, tgt, n, cs, cd, bf16)
        _, t_plan = hip_timed(plans)
        for name, t in (("insert_ms", t_ins), ("kernel_map_ms", t_km), ("plans_ms", t_plan)):
            best[name] = min(best.get(name, 1e9), t)
    key_bytes = 4 * (D + 1)
    out.update({k: round(v, 4) for k, v in best.items()})
    probe_bytes = n * K * key_bytes + 8 * n_pairs
    out["kernel_map_GBs"] = round(probe_bytes / (best["kernel_map_ms"] * 1e-3) / 1e9, 1)
    out["insert_GBs"] = round((n * key_bytes + 8 * 2 * n + 20 * n) / (best["insert_ms"] * 1e-3) / 1e9, 1)
    out["kernel_map_frac_of_hbm_peak"] = round(out["kernel_map_GBs"] / PEAK_HBM_GBS, 4)
    out["note"] = ("kernel_map_ms = end-to-end build (probe + offsets + compaction) as one stream of launches; bytes = "
                   "N*K*4(D+1) probes + 8 B per pair written (SURVEY 8d); built once per layer geometry and cached")
    return out


def rank_points(n, rank, world, imbalance):
    """voxels of this rank's scene: n, or with --imbalance a ramp from 0.75 n (rank 0) to 1.25 n (last rank)"""
    if not imbalance or world < 2:
        return n
    return int(n * (0.75 + 0.5 * rank / (world - 1)))


def make_scene(n, extent, seed, D=3):
    """unique, unsorted voxels drawn uniformly from [0, extent)^D (extent: int or one
    value per axis), batch index 0."""
    g = torch.Generator().manual_seed(seed)
    ext = [extent] * D if isinstance(extent, int) else list(extent)
    pts = torch.stack([torch.randint(0, e, (int(1.6 * n),), generator=g) for e in ext], 1) if len(set(ext)) > 1 \
        else torch.randint(0, ext[0], (int(1.6 * n), D), generator=g)
    pts = torch.unique(pts, dim=0)
    pts = pts[torch.randperm(pts.shape[0], generator=g)][:n]
    assert pts.shape[0] == n, "extent too small for n unique voxels"
    return torch.cat([torch.zeros(n, 1, dtype=torch.long), pts], 1).int().contiguous()


REFERENCE_THREAD_CAP = 16   # the reference caps its own OpenMP threads (MinkowskiEngine/__init__.py:34-46)


class _threads:
    """with _threads(n): torch / MKL / OpenMP thread count for the CPU baseline (n = 0: leave as is)"""
    def __init__(self, n):
        self.n, self.old = n, None

    def __enter__(self):
        if self.n:
            self.old = torch.get_num_threads()
            torch.set_num_threads(self.n)

    def __exit__(self, *exc):
        if self.old:
            torch.set_num_threads(self.old)
        return False


def cpu_baseline_both(fn, budget_s):
    """`fn(budget)` timed at the box's full thread count (the reported baseline: the most favourable for the reference)
    and, beside it, under the reference's own 16-thread cap when the box has more cores."""
    full = fn(budget_s)
    if full is not None and full.get("kind") == "reference" and torch.get_num_threads() > REFERENCE_THREAD_CAP:
        with _threads(REFERENCE_THREAD_CAP):
            capped = fn(max(1.0, budget_s / 3))
        if capped is not None:
            full["capped"] = {k: capped[k] for k in ("value", "unit", "cores", "ms_per_step") if k in capped}
            full["capped"]["note"] = "same sample under the reference's own thread cap (MinkowskiEngine/__init__.py:34-46)"
    return full


def cpu_baseline(coords, feats, kernel, budget_s):
    """The reference's CPU path (ConvolutionForwardCPU / ConvolutionBackwardCPU of oracle/_ref/_C.so,
    built unmodified from the reference sources) on the same scene; falls back to the numpy port."""
    n = coords.shape[0]
    D = coords.shape[1] - 1
    cin, cout = kernel.shape[1:]
    cores = torch.get_num_threads()
    try:
        from oracle import ref
        if not ref.available():
            raise RuntimeError("oracle/_ref/_C.so not present")
        rc = ref.RefConv(coords, 3)
        y = rc.forward(feats, kernel)          # builds the kernel map (cached afterwards)
        gy = torch.ones_like(y)
        rc.backward(feats, gy, kernel)
        times, t_end = [], time.perf_counter() + budget_s
        while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 50):
            t0 = time.perf_counter()
            rc.forward(feats, kernel)
            rc.backward(feats, gy, kernel)
            times.append(time.perf_counter() - t0)
        best = min(times)
        return {"value": round(n / best / 1e6, 4), "unit": "Mpoints/s", "cores": cores, "kind": "reference",
                "sample": f"full workload ({n} voxels, {D}-D, {cin}->{cout}, k=3, kernel map cached), min of "
                          f"{len(times)} fwd+bwd iterations, MKL sgemm via libtorch, {cores} threads",
                "ms_per_step": round(best * 1e3, 3)}
    except Exception as e:  # noqa: BLE001
        from oracle import me_oracle as O
        import numpy as np
        m = min(n, 20000)
        co = coords[:m].numpy()
        _, km = O.kernel_map(co, co, O.make_region(D, 3))
        x, w = feats[:m].numpy(), kernel.numpy()
        t0 = time.perf_counter()
        y = O.conv_forward(x, w, km, m, dtype=np.float32)
        O.conv_backward(x, np.ones_like(y), w, km, dtype=np.float32)
        dt = time.perf_counter() - t0
        return {"value": round(m / dt / 1e6, 4), "unit": "Mpoints/s", "cores": cores, "kind": "port",
                "sample": f"first {m} voxels of the workload, one fwd+bwd, numpy port ({e})"}


def cpu_baseline_minkunet(coords, layer_specs, budget_s):
    """The convolution layers of the network (recorded from the GPU run) on the reference's own CPU operators on
    the same scene (oracle/ref.py RefConvStack); batch norm / ReLU are not replayed, so this is a LOWER bound of
    the reference's step time — bounded to about `budget_s` seconds."""
    from oracle import ref
    if not ref.available():
        return None
    n = coords.shape[0]
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    stack = ref.RefConvStack(coords, layer_specs)
    stack.run()                                    # builds the coordinate and kernel maps (cached afterwards)
    t_first = time.perf_counter() - t0
    times = []
    t_end = time.perf_counter() + budget_s
    while not times or (time.perf_counter() + min(times) < t_end and len(times) < 5):
        t0 = time.perf_counter()
        stack.run()
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": round(n / best / 1e6, 5), "unit": "Mpoints/s", "cores": cores, "kind": "reference",
            "ms_per_step": round(best * 1e3, 1), "first_pass_ms": round(t_first * 1e3, 1),
            "sample": f"the {len(layer_specs)} convolution layers of MinkUNet34C (forward + backward, maps cached) on "
                      f"the same {n}-voxel scene through the reference's ConvolutionForward/BackwardCPU (+ transposed "
                      f"twins; 1x1 layers as its use_mm matmuls), min of {len(times)} passes, {cores} threads; batch "
                      "norm / ReLU / loss / optimizer NOT included (a lower bound of the reference's step)"}


def kernel_table(timer, steps):
    """Per hot kernel: launches per step, mean ms per launch, algorithmic TFLOP/s over all its launches."""
    out = {}
    for name, (launches, total_ms, flops) in timer.totals().items():
        out[name] = {"launches_per_step": round(launches / steps, 2), "avg_ms": round(total_ms / launches, 4),
                     "ms_per_step": round(total_ms / steps, 4),
                     "tflops": round(flops / (total_ms * 1e-3) / 1e12, 2) if total_ms > 0 else None}
    return out


def set_kernel_timer(MEB, timer):
    """per-launch HIP-event timing on: the python host's KernelTimer hook and the native host layer's own recorder"""
    from minkowskiengine_amd import host
    MEB.KERNEL_TIMER = timer
    if host.native_module() is not None:
        host.native_module().timing_enable(timer is not None)


LAST_RUN = {}    # side results of the last run_timed call: per-rank times of the reported block


class Exchange:
    """The data-parallel gradient exchange of a bench step, two ways (--exchange):
      arena  (default) distributed.GradientArena: gradients are born in one flat buffer, ONE RCCL all-reduce after the
             backward pass — no per-parameter hooks, no bucket copies (torch DDP costs the host-bound MinkUNet34C step
             1.6 ms on ONE rank where the all-reduce itself is free: scripts/ddp_overhead.py);
      ddp    torch DistributedDataParallel, 25 MB buckets overlapped with the backward pass (the reference's recipe,
             examples/multigpu_ddp.py:81-95).
    Without a process group both are the plain module."""

    def __init__(self, module, dev, dist_utils, mode, sync_bn=False, chunks=1):
        self.dist, self.mode, self.module = dist_utils, mode, module
        self.active = dist_utils.exchange_active()
        self.arena, self.net = None, module
        if not self.active:
            self.mode = "none"
        elif mode == "ddp":
            self.net = dist_utils.data_parallel(module, dev, sync_batchnorm=sync_bn)
        else:
            if sync_bn:
                import minkowskiengine_amd as ME
                self.net = self.module = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(module)
            self.arena = dist_utils.GradientArena(self.module, chunks=chunks)

    def zero_grad(self):
        if self.arena is not None:
            self.arena.zero_grad()
        else:
            for p in self.module.parameters():
                p.grad = None

    def backward(self, fn, sync=True):
        """run fn() (the backward pass) with (sync) or without the exchange"""
        if self.mode == "ddp" and not sync:
            with self.dist.no_sync(self.net):
                fn()
            return
        fn()
        if self.arena is not None and sync:
            self.arena.all_reduce()

    def describe(self):
        if self.mode == "ddp":
            return f"torch DDP over {self.dist.backend_name()} (25 MB gradient buckets overlapped with backward)"
        if self.mode == "arena":
            return (f"gradient arena over {self.dist.backend_name()} (gradients born in one flat buffer, one all-reduce "
                    "per step after backward)")
        return ""


def gpu_state_under_load(step, min_steps=4):
    """Clock / power / temperature of GPU 0 sampled by rocm-smi WHILE `step` keeps the device busy (a sample taken after the
    timed region would show the idle clocks).  -> dict or None (no rocm-smi, unparsable output)."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    try:
        p = subprocess.Popen([exe, "-d", "0", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        n = 0
        t_end = time.perf_counter() + 20.0
        while (p.poll() is None or n < min_steps) and time.perf_counter() < t_end:
            step()
            n += 1
        torch.cuda.synchronize()
        out, _ = p.communicate(timeout=10)
        card = next(iter(json.loads(out).values()))
    except Exception:  # noqa: BLE001
        return None
    pick = {}
    for k, v in card.items():
        kl = k.lower()
        if "sclk" in kl or "mclk" in kl or "fclk" in kl or "power" in kl or "performance level" in kl or \
                ("temperature" in kl and ("edge" in kl or "junction" in kl or "hotspot" in kl)):
            pick[k] = v
    pick["steps_during_sample"] = n
    return pick or None


def timed_block(step, steps, dist_utils, dev, reps=3):
    """median over `reps` of [barrier + synchronize | `steps` steps | synchronize + barrier], max over ranks -> seconds"""
    out = []
    for _ in range(reps):
        dist_utils.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dist_utils.barrier()
        out.append(dist_utils.max_over_ranks(time.perf_counter() - t0, dev))
    return sorted(out)[(len(out) - 1) // 2]


def allreduce_probe(n_bytes_total, dist_utils, dev, dtype=torch.float32, bucket_bytes=25 * 1024 * 1024, reps=5):
    """One gradient exchange on its own: the model's gradient bytes all-reduced in DDP-sized flat buckets with nothing
    else on the device, HIP-event timed on the collective's stream of record (the current stream waits for it) ->
    (ms for all buckets [max over ranks], bucket count).  The reference point for what the step could hide."""
    import torch.distributed as dist
    esz = torch.empty(0, dtype=dtype).element_size()
    sizes, left = [], n_bytes_total
    while left > 0:
        sizes.append(min(left, bucket_bytes))
        left -= sizes[-1]
    bufs = [torch.zeros(max(1, b // esz), dtype=dtype, device=dev) for b in sizes]
    best = None
    for _ in range(reps + 1):
        dist_utils.barrier()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for b in bufs:
            dist.all_reduce(b)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e)
        best = t if best is None else min(best, t)
    return dist_utils.max_over_ranks(best, dev), len(sizes)


def run_timed(step, args, dist_utils, MEB, dev, timers_in_blocks=True):
    """-> (seconds of the MEDIAN K-step block [max over ranks], every block's seconds, KernelTimer of ALL timed
    blocks, number of timed steps).  Each block: barrier + synchronize | K steps | synchronize + barrier.
    timers_in_blocks=False (host-bound loops: a new scene every step): the per-launch HIP events — two event records
    per convolution launch, ~1 ms of host time per MinkUNet step — stay out of the timed blocks and the kernel table
    comes from ONE extra, untimed block."""
    for _ in range(args.warmup):
        step()
    timer = MEB.KernelTimer()
    timer._native_records()                       # (drop what an earlier, unrelated pass left in the native recorder)
    timer.native.clear()
    timer.flops.clear()
    if not timers_in_blocks:
        torch.cuda.synchronize()
        set_kernel_timer(MEB, timer)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        set_kernel_timer(MEB, None)
    blocks, local = [], []
    total = 0.0
    n_timed_blocks = 0
    while True:
        dist_utils.barrier()
        torch.cuda.synchronize()
        # per-launch HIP events (two event records per hot kernel) in every `timer_blocks`-th block of the timed region
        timed_block_now = timers_in_blocks and len(blocks) % max(1, args.timer_blocks) == 0
        set_kernel_timer(MEB, timer if timed_block_now else None)
        n_timed_blocks += 1 if timed_block_now else 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0                         # this rank's own K steps (before it waits for the others)
        dist_utils.barrier()
        elapsed = time.perf_counter() - t0
        set_kernel_timer(MEB, None)
        elapsed = dist_utils.max_over_ranks(elapsed, dev)      # the same value on every rank: same loop exit
        blocks.append(elapsed)
        local.append(own)
        total += elapsed
        if (total >= args.min_time and len(blocks) >= args.min_blocks) or len(blocks) >= args.max_blocks:
            break
    srt = sorted(blocks)
    median = srt[(len(srt) - 1) // 2]      # lower median: a measured block, never an interpolation
    LAST_RUN["per_rank_ms_per_step"] = [round(v / args.steps * 1e3, 4) for v in
                                        dist_utils.gather_over_ranks(local[blocks.index(median)], dev)]
    every = max(1, args.timer_blocks)
    with_ev = [b for i, b in enumerate(blocks) if timers_in_blocks and i % every == 0]
    without = [b for i, b in enumerate(blocks) if not (timers_in_blocks and i % every == 0)]
    med = lambda v: sorted(v)[(len(v) - 1) // 2] / args.steps * 1e3 if v else None   # noqa: E731
    LAST_RUN["event_blocks"] = {"blocks_with_kernel_events": len(with_ev), "blocks_without": len(without),
                                "median_ms_per_step_with_events": round(med(with_ev), 4) if with_ev else None,
                                "median_ms_per_step_without": round(med(without), 4) if without else None}
    return median, blocks, timer, args.steps * (n_timed_blocks if timers_in_blocks else 1)


def roofline_entry(kernel, flops, compulsory_bytes, avg_ms, bf16, traffic, traffic_src, split=False):
    """the roof is chosen by the launch's arithmetic intensity on its COMPULSORY bytes.
    split: an fp32 workload computed by k_conv_tile_f32x3 — `peak` / `frac` are those of the pipe the kernel issues to
    (six bf16 MFMAs per fp32 product block: 2500 / 6 TFLOP/s); the fraction of the fp32 MFMA peak (the denominator of
    rounds 1 and 2) is reported next to it."""
    peak_t = PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_F32_MATRIX_TFLOPS
    ridge = RIDGE_BF16 if bf16 else RIDGE_F32
    intensity = flops / compulsory_bytes
    tflops = flops / (avg_ms * 1e-3) / 1e12
    gbs = compulsory_bytes / (avg_ms * 1e-3) / 1e9
    r = {"kernel": kernel, "flops_per_launch": flops, "compulsory_bytes_per_launch": int(compulsory_bytes),
         "intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
         "avg_ms": round(avg_ms, 4), "tflops": round(tflops, 2), "frac_of_mfma_peak": round(tflops / peak_t, 4),
         "compulsory_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4)}
    if intensity >= ridge:
        r.update({"bound": "mfma", "achieved": round(tflops, 2), "peak": peak_t, "unit": "TFLOP/s",
                  "frac": round(tflops / peak_t, 4)})
    else:
        r.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                  "frac": round(gbs / PEAK_HBM_GBS, 4)})
    if split:
        # the kernel issues to the bf16 matrix pipe, six MFMAs per fp32 product block: THAT pipe's ceiling for this
        # arithmetic (2500 / 6 TFLOP/s of fp32-grade products) is the roof `frac` is quoted against; the fraction of
        # the fp32-MFMA peak (the round-1 / round-2 denominator, which this kernel can exceed) stays beside it
        pipe_peak = PEAK_BF16_MATRIX_TFLOPS / 6.0
        r["pipe"] = ("bf16 matrix pipe: fp32 operands split exactly into three bf16 terms, six v_mfma_f32_16x16x32_bf16 "
                     "per product block, fp32 accumulation (fp32-grade results, tests/test_gpu_conv.py)")
        r["peak_of_pipe"] = round(pipe_peak, 1)
        r["frac_of_pipe"] = round(tflops / pipe_peak, 4)
        r["frac_of_fp32_mfma_peak"] = round(tflops / PEAK_F32_MATRIX_TFLOPS, 4)
        if r["bound"] == "mfma":
            r.update({"peak": round(pipe_peak, 1), "frac": r["frac_of_pipe"]})
    r["traffic"] = traffic
    r["traffic_note"] = ("HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE) — a CITED constant from the committed "
                         f"rocprofv3 --pmc passes ({traffic_src}); replaced by an in-run measurement when the run's PMC "
                         "budget allows (--pmc)" if traffic is not None else
                         "no committed PMC pass for this workload; measured in-run when the PMC budget allows (--pmc)")
    # (kernel-name fragments that select this launch in a rocprofv3 counter table: attach_traffic)
    m = kernel.split("<")
    r["pmc_match"] = [m[0].strip() + "<" + m[1].split(">")[0].replace(",", ", ") + ","] if len(m) > 1 else [m[0].strip()]
    return r


def bench_conv(args, ME, MEB, dist_utils, rank, world, dev, startup):
    if args.workload == "conv4d":
        D, n, extent, cin, cout = 4, args.points or 400000, (100, 100, 100, 8), args.cin or 32, args.cout or 64
        cfg, ext_s = "BASELINE configs[4]", "[0,100)^3 x [0,8)"
    else:
        D, n, extent, cin, cout = 3, args.points or 100000, args.extent, args.cin or 64, args.cout or 128
        cfg, ext_s = "BASELINE configs[1]", f"[0,{args.extent})^3"
    K = 3 ** D
    n = rank_points(n, rank, world, args.imbalance)
    coords = make_scene(n, extent, seed=rank, D=D)            # one independent scene per rank
    g = torch.Generator().manual_seed(1000 + rank)
    feats = torch.rand(n, cin, generator=g)
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, dimension=D, bias=False).to(dev)
    ex = Exchange(conv, dev, dist_utils, args.exchange)       # broadcasts rank 0's weights when a group exists
    net = ex.net

    # cold path: coordinate insertion + kernel map + tile plans + first forward/backward, in pieces
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    x = ME.SparseTensor(feats.to(dev).to(tdt), coords.to(dev), requires_grad=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    y = net(x)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    y.F.sum().backward()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    cold_ms = (t3 - t0) * 1e3
    mgr_ = x.coordinate_manager._manager
    if ME.is_native():
        n_pairs = int(mgr_.kernel_map_pairs(x.coordinate_map_key, y.coordinate_map_key, [3] * D, [1] * D, [1] * D, 0,
                                            False, False))
    else:
        n_pairs = mgr_._kernel_map(x.coordinate_map_key, y.coordinate_map_key, [3] * D, [1] * D, [1] * D,
                                   ME.RegionType.HYPER_CUBE, None, False, False).n_pairs
    grad_seed = torch.ones_like(y.F)

    def step(zero=True, sync=True):
        if zero:
            ex.zero_grad()
        x.F.grad = None
        out = net(x)
        ex.backward(lambda: out.F.backward(grad_seed), sync)

    if getattr(args, "pmc_child_mode", False):     # profiled child (measure_traffic): the steps and nothing else
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        return None
    best, blocks, timer, timed_steps = run_timed(step, args, dist_utils, MEB, dev)
    per_rank = LAST_RUN.get("per_rank_ms_per_step")
    total_points = dist_utils.sum_over_ranks(n, dev)
    pairs_all = dist_utils.sum_over_ranks(n_pairs, dev)
    multi = None
    if dist_utils.exchange_active():
        # What the one-layer headline hides nothing behind: its 0.88 MB all-reduce starts when the ONLY layer's backward
        # is done.  Beside the synchronous step: the same step without the exchange (DDP.no_sync), the exchange on its
        # own, and a gradient-accumulation window (A - 1 micro-steps under no_sync, the A-th reduces) — the loop a
        # one-layer data-parallel job would actually run.
        def step_nosync(zero=True):
            step(zero, sync=False)
        A = max(2, args.accum)

        def window():
            for i in range(A):                      # gradients accumulate over the window, the last micro-step reduces
                if i + 1 < A:
                    step_nosync(zero=(i == 0))
                else:
                    step(zero=False)
        t_nosync = timed_block(step_nosync, args.steps, dist_utils, dev)
        t_window = timed_block(window, max(1, args.steps // A), dist_utils, dev)
        ar_ms, n_buckets = allreduce_probe(conv.kernel.numel() * 4, dist_utils, dev)
        ms_sync = best / args.steps * 1e3
        ms_nosync = t_nosync / args.steps * 1e3
        ms_window = t_window / (max(1, args.steps // A) * A) * 1e3
        multi = dict(dist_utils.collective_info(), exchange=ex.mode,
                     arena=(ex.arena.describe() if ex.arena is not None else None),
                     arena_per_rank=arena_per_rank(ex, dist_utils, dev),
                     per_rank_ms_per_step=per_rank,
                     no_sync_ms_per_step=round(ms_nosync, 4),
                     allreduce_ms={"standalone": round(ar_ms, 4), "buckets": n_buckets,
                                   "bytes": int(conv.kernel.numel() * 4),
                                   "exposed_in_step": round(ms_sync - ms_nosync, 4)},
                     parallel_efficiency=round(ms_nosync / ms_sync, 4),
                     accumulation={"window": A, "ms_per_micro_step": round(ms_window, 4),
                                   "value": round(total_points / (ms_window * 1e-3) / 1e6, 3),
                                   "parallel_efficiency": round(ms_nosync / ms_window, 4),
                                   "note": f"{A - 1} micro-steps under DDP.no_sync, the {A}-th all-reduces the accumulated "
                                           "gradient: value = voxels of all ranks per micro-step time"},
                     note="parallel_efficiency = step time without the gradient exchange (DDP.no_sync, max over ranks) / "
                          "step time with it, on THIS world size — the share of the step that is not exposed exchange; "
                          "the 1 -> N scaling efficiency is the driver's to compute from the per-N values")
    cold = cold_path(ME, MEB, feats, coords.to(dev), dev, n, D, K, cin, cout, args.dtype == "bf16") if rank == 0 else None
    if rank != 0:
        return None
    kernels = kernel_table(timer, timed_steps)
    flops_per_launch = 2.0 * n_pairs * cin * cout     # each of forward / dgrad / wgrad
    nc = 32 if 0 < cout % 64 <= 32 else 64                     # conv_variant() of csrc/conv.hip
    kc = 64 if cin % 64 == 0 else 32 if cin % 32 == 0 else 16
    bf16 = args.dtype == "bf16"
    if bf16:
        # conv_variant_bf16() of csrc/conv_bf16.hip
        kc = 128 if cin % 128 == 0 else 96 if cin % 96 == 0 else 32 if cin <= 32 else 64 if cin <= 64 else 128
    from minkowskiengine_amd import _lib
    split = (not bf16) and MEB._use_split(_lib.load(), cin, cout)   # (same policy in csrc_host/manager.cpp use_split)
    if split:
        # conv_variant_f32x3() of csrc/conv_f32x3.hip: fp32 operands split exactly into three bf16 terms, six bf16 MFMAs
        nc = 128 if cout % 128 == 0 else 96 if cout % 96 == 0 else 32 if cout <= 32 else 64
        kc = 128 if cin % 128 == 0 and nc <= 64 else 64 if cin % 64 == 0 else 96 if cin % 96 == 0 else \
            32 if cin <= 32 else 64
    # (64- and 128-column slabs of the split kernel run its wave-specialised instantiation)
    kname = "k_conv_tile_bf16" if bf16 else ("k_conv_tile_f32x3_ws" if nc in (64, 128) else "k_conv_tile_f32x3") \
        if split else "k_conv_tile_f32"
    esz = 2 if bf16 else 4
    compulsory = esz * (n * cin + n * cout + K * cin * cout) + 8 * n_pairs   # SURVEY 8d, forward
    traffic, traffic_src = pmc_traffic("k_conv_tile_bf16_forward" if bf16 else kname, n, extent, cin,
                                       cout) if D == 3 else (None, None)
    ms_step = best / args.steps * 1e3
    line = {
        "metric": f"MinkowskiConvolution fwd+bwd Mpoints/sec ({n // 1000}k-pt {D}D, k=3)",
        "value": round(total_points / (best / args.steps) / 1e6, 3),
        "unit": "Mpoints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"single MinkowskiConvolution {D}D k=3 s=1, {n} voxels/GPU uniform in "
                               f"{ext_s}, {cin}->{cout} ch, {'bf16 features / fp32 accumulate' if bf16 else 'fp32'}, "
                               f"kernel map cached ({cfg})",
                   "host_layer": ME.get_host(),
                   "points_per_gpu": n, "pairs_per_gpu": n_pairs, "pairs_total": int(pairs_all),
                   "parallelism": f"scene-sharded dp{world}" + (", " + ex.describe() if dist_utils.exchange_active() else ""),
                   "imbalance": bool(args.imbalance and world > 1),
                   "oversubscribed": world > max(1, dist_utils.visible_gpus())},
        "multi_gpu": multi,
        "timing": {"blocks": len(blocks), "steps_per_block": args.steps,
                   "blocks_ms_per_step": [round(b / args.steps * 1e3, 4) for b in blocks],
                   "timed_region_s": round(sum(blocks), 4),
                   "fastest_block_ms_per_step": round(min(blocks) / args.steps * 1e3, 4),
                   "kernel_events": LAST_RUN.get("event_blocks"),
                   "reported": "median block (max over ranks inside each block); the per-launch HIP events behind "
                               "`roofline` / `kernels` are recorded in every --timer-blocks-th block of the timed region"},
        "roofline": roofline_entry(f"{kname}<{nc},{kc}> (forward)", flops_per_launch,
                                   compulsory, kernels["conv_forward"]["avg_ms"], bf16, traffic, traffic_src, split),
        "kernels": kernels,
        "cold_ms": round(cold_ms, 2),
        "cold_breakdown_ms": {"process_startup_to_first_launch": startup,
                              "sparse_tensor_insert": round((t1 - t0) * 1e3, 2),
                              "first_forward_kernel_map_plan": round((t2 - t1) * 1e3, 2),
                              "first_backward_plans": round((t3 - t2) * 1e3, 2),
                              "note": "cold_ms = sparse tensor + first forward + first backward on this process' FIRST "
                                      "use of the library; it contains the one-time load of the code object's kernels "
                                      "(hipModule lazy loading, reported under process_startup_to_first_launch."
                                      "first_kernel_ms for one trivial kernel), the allocator's first hipMallocs and the "
                                      "host synchronisations of the build; the warm build is under `cold`"},
        "cold": cold,
    }
    # compulsory HBM bytes of each pass (every tensor touched once, the pair lists read once): forward = X + Y +
    # W + 8P; input gradient = dY + dX + W + 8P; weight gradient = X + dY + dW + 8P (its reduce: the partial images are
    # not compulsory — dW once)
    line["roofline"]["compulsory_by_pass"] = {
        "forward": int(compulsory),
        "dgrad": int(esz * (n * cout + n * cin + K * cin * cout) + 8 * n_pairs),
        "wgrad": int(esz * (n * cin + n * cout) + 4 * K * cin * cout + 8 * n_pairs),
        "wgrad_reduce": int(4 * K * cin * cout)}
    if world == 1 and args.cpu_budget > 0:      # the reference CPU path is fp32 whatever our feature dtype
        w_cpu = conv.kernel.detach().float().cpu()
        line["cpu_baseline"] = cpu_baseline_both(lambda b: cpu_baseline(coords, feats, w_cpu, b), args.cpu_budget)
        line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
    else:
        line["cpu_baseline"] = None
    return line


def bench_minkunet(args, ME, MEB, dist_utils, rank, world, dev, startup):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import minkunet as MU
    n = rank_points(args.points or 200000, rank, world, args.imbalance)
    coords = MU.synthetic_scene(n, seed=rank)
    n = coords.shape[0]
    g = torch.Generator().manual_seed(1000 + rank)
    feats = torch.rand(n, 3, generator=g)
    torch.manual_seed(0)
    model = MU.MinkUNet34C(3, 20, D=3).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    ex = Exchange(model, dev, dist_utils, args.exchange, sync_bn=args.sync_bn, chunks=args.arena_chunks)
    net = ex.net
    # (torch's fused multi-tensor SGD: one launch for all 188 parameters — 0.23 ms against 0.34 ms for the foreach form,
    # scripts/sgd_step_time.py; same arithmetic)
    try:
        opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, fused=True)
    except (TypeError, RuntimeError, ValueError):
        opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    labels = torch.randint(0, 20, (n,), generator=g).to(dev)
    crit = torch.nn.CrossEntropyLoss() if args.torch_loss else MU.cross_entropy   # same value and gradient
    bf16 = args.dtype == "bf16"
    tdt = torch.bfloat16 if bf16 else torch.float32
    x = ME.SparseTensor(feats.to(dev).to(tdt), coords.to(dev))   # coordinate + kernel maps cached in x's manager

    # layer census for the CPU baseline, recorded by forward hooks during the first step
    specs, hooks = [], []
    if rank == 0 and world == 1 and args.cpu_budget > 0:
        def record(mod, inp, out):
            kg = mod.kernel_generator
            specs.append((bool(mod.is_transpose), list(kg.kernel_size), list(kg.kernel_stride), mod.in_channels,
                          mod.out_channels, list(inp[0].tensor_stride)))
        hooks = [m.register_forward_hook(record) for m in model.modules()
                 if isinstance(m, (ME.MinkowskiConvolution, ME.MinkowskiConvolutionTranspose))]

    feats_dev, coords_dev = x.F.detach(), coords.to(dev)
    pending = [None]
    if args.scenes == "fresh":
        # the library's recipe replay, as a training loop over new scenes would set it: each new manager rebuilds what
        # the previous scene's network asked for in one burst (strided maps, kernel maps, then ALL tile plans in four
        # launches: me_plan_build_multi) instead of lazily, layer by layer
        ME.set_map_prefetch(bool(args.replay_maps))
    if args.scenes == "pipelined":
        # a new scene every step, its maps built by a loader thread on a side stream while this thread launches the
        # previous step (ME.utils.ScenePrefetcher: the previous scene's build requests are replayed right after the
        # coordinate insert, all tile plans in four launches; the native host releases the GIL inside the builds)
        def scenes_forever():
            while True:
                yield feats_dev, coords_dev
        loader = ME.utils.ScenePrefetcher(scenes_forever(), depth=args.loader_depth)
        pending[0] = iter(loader)

    def step(sync=True):
        ex.zero_grad()
        if args.scenes == "cached":
            xin = x
        elif args.scenes == "fresh":
            xin = ME.SparseTensor(feats_dev, coords_dev)      # coordinate maps, kernel maps, plans rebuilt lazily
        else:
            xin = next(pending[0])
            if os.environ.get("ME_BENCH_DISCARD_LOADED") == "1":   # (interference experiment: the loader runs, the step
                xin = x                                            # trains on the cached scene)
        loss = crit(net(xin).F.float(), labels)   # mean cross-entropy over the voxels
        ex.backward(loss.backward, sync)
        opt.step()

    if getattr(args, "pmc_child_mode", False):     # profiled child (measure_traffic): exactly `steps` steps
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        return None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t0) * 1e3
    for h in hooks:
        h.remove()
    graphed = False
    if args.graph and args.scenes != "cached" and os.environ.get("ME_BENCH_DISCARD_LOADED") != "1":
        raise SystemExit("--graph needs --scenes cached (a captured step replays fixed shapes and addresses)")
    if args.graph:
        step, graphed = capture_step(step), True
    # (per-launch HIP events — ~130 event pairs per step — stay out of the timed blocks: the kernel table comes from ONE
    # extra, untimed block, so that `ms_per_step` is the step a training loop would see)
    best, blocks, timer, timed_steps = run_timed(step, args, dist_utils, MEB, dev, timers_in_blocks=False)
    per_rank = LAST_RUN.get("per_rank_ms_per_step")
    total_points = dist_utils.sum_over_ranks(n, dev)
    multi = None
    if dist_utils.exchange_active():
        # BASELINE configs[3]: what the exchange costs THIS step — the step without it (DDP.no_sync: same kernels, no
        # collectives), the 151 MB of gradients all-reduced on their own, and the difference the bucketed overlap leaves
        def step_nosync():
            step(sync=False)
        t_nosync = timed_block(step_nosync, args.steps, dist_utils, dev)
        grad_bytes = sum(p.numel() * p.element_size() for p in model.parameters() if p.requires_grad)
        ar_ms, n_buckets = allreduce_probe(grad_bytes, dist_utils, dev)
        ms_sync, ms_nosync = best / args.steps * 1e3, t_nosync / args.steps * 1e3
        multi = dict(dist_utils.collective_info(), exchange=ex.mode,
                     arena=(ex.arena.describe() if ex.arena is not None else None),
                     arena_per_rank=arena_per_rank(ex, dist_utils, dev),
                     per_rank_ms_per_step=per_rank,
                     per_rank_points=[int(v) for v in dist_utils.gather_over_ranks(n, dev)],
                     no_sync_ms_per_step=round(ms_nosync, 3),
                     allreduce_ms={"standalone": round(ar_ms, 3), "buckets": n_buckets, "bytes": int(grad_bytes),
                                   "exposed_in_step": round(ms_sync - ms_nosync, 3),
                                   "hidden_by_overlap": round(max(0.0, ar_ms - (ms_sync - ms_nosync)), 3)},
                     parallel_efficiency=round(ms_nosync / ms_sync, 4),
                     sync_bn=bool(args.sync_bn), imbalance=bool(args.imbalance),
                     note="parallel_efficiency = step time without the gradient exchange (DDP.no_sync, max over ranks) / "
                          "step time with it on THIS world size; allreduce_ms.standalone = all gradient buckets reduced "
                          "with nothing else running (HIP events), exposed_in_step = what the overlapped step still pays")
    # the same step replayed from a hipGraph (maps cached: fixed shapes and addresses) — the GPU time of the step with
    # the host out of the way; reported beside the eager figure, never instead of it
    graph_ms, graph_blocks = None, []
    if args.scenes == "cached" and not graphed and world == 1 and not args.no_graph_probe and not dist_utils.exchange_active():
        try:
            replay = capture_step(step)
            for _ in range(2):
                replay()
            graph_blocks = []
            for _ in range(max(3, min(len(blocks), 9))):      # as many blocks as the eager measurement took (3 .. 9)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    replay()
                torch.cuda.synchronize()
                graph_blocks.append((time.perf_counter() - t0) / args.steps * 1e3)
            graph_ms = sorted(graph_blocks)[(len(graph_blocks) - 1) // 2]
        except Exception as e:  # noqa: BLE001
            graph_ms = f"capture failed: {type(e).__name__}: {e}"
    if rank != 0:
        return None
    kernels = kernel_table(timer, timed_steps)
    tot_ms = sum(k["ms_per_step"] for k in kernels.values())
    tot_flops = sum(f for _, _, f in timer.totals().values()) / max(timed_steps, 1)
    peak = PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_F32_MATRIX_TFLOPS
    ms_step = best / args.steps * 1e3
    achieved = round(tot_flops / (tot_ms * 1e-3) / 1e12, 2) if tot_ms > 0 else None
    whole = round(1567e9 / (ms_step * 1e-3) / 1e12, 2)
    line = {
        "metric": "MinkUNet34C fwd+bwd+SGD Mpoints/sec (200k-pt synthetic scene)",
        "value": round(total_points / (best / args.steps) / 1e6, 3),
        "unit": "Mpoints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"MinkUNet34C (3 -> 20 classes, {n_params} parameters) "
                               f"forward + cross-entropy + backward + SGD step, {n} voxels/GPU on a union of 9 planes "
                               f"in 400^3 (SURVEY 8d), {'bf16 activations / fp32 master weights and accumulation' if bf16 else 'fp32'}, "
                               + {"cached": "maps cached", "fresh": "a NEW scene every step (all maps and plans rebuilt" + (", recipe replay)" if args.replay_maps
                                                                                                    else ", lazily)"),
                                  "pipelined": "a NEW scene every step, its maps built by a loader thread on a side stream "
                                               "during the previous step (ME.utils.ScenePrefetcher)"}[args.scenes]
                               + " (BASELINE configs[2]; configs[3] with N = 8)",
                   "scenes": args.scenes,
                   "map_prefetch": bool(ME.map_prefetch_enabled()),
                   "loader": ({"depth": args.loader_depth,
                               "build_ms_median": round(sorted(loader.build_ms)[len(loader.build_ms) // 2], 3),
                               "consumer_wait_ms_median": round(sorted(loader.wait_ms)[len(loader.wait_ms) // 2], 3)}
                              if args.scenes == "pipelined" else None),
                   "host_layer": ME.get_host(),
                   "points_per_gpu": n,
                   "parallelism": f"scene-sharded dp{world}" + (
                       ", " + ex.describe() + (", MinkowskiSyncBatchNorm" if args.sync_bn else ", per-rank batch norm")
                       if dist_utils.exchange_active() else ""),
                   "hip_graph": graphed,
                   "imbalance": bool(args.imbalance and world > 1),
                   "oversubscribed": world > max(1, dist_utils.visible_gpus())},
        "multi_gpu": multi,
        "timing": {"blocks": len(blocks), "steps_per_block": args.steps,
                   "blocks_ms_per_step": [round(b / args.steps * 1e3, 3) for b in blocks],
                   "timed_region_s": round(sum(blocks), 4),
                   "fastest_block_ms_per_step": round(min(blocks) / args.steps * 1e3, 3),
                   "reported": "median block (max over ranks inside each block); per-kernel HIP events recorded in one "
                               "extra block outside the timed region"},
        "roofline": {"bound": "mfma", "kernel": "all convolution launches of a step (k_conv_tile_* / k_conv_halo / k_conv_stem forward + dgrad, "
                                                "k_wgrad_*); HIP-event timed" + (" in a separate eager pass: the timed "
                                                "region replays a hipGraph" if graphed else ""),
                     "achieved": achieved if achieved is not None else whole,
                     "peak": peak, "unit": "TFLOP/s",
                     "frac": round((achieved if achieved is not None else whole) / peak, 4),
                     "whole_step_tflops": whole, "whole_step_frac": round(whole / peak, 4),
                     "traffic": None, "pmc_match": None,
                     "traffic_note": "HBM-side bytes per step of all convolution launches: measured in-run when the PMC "
                                     "budget allows (--pmc)",
                     "flops_per_step": tot_flops, "conv_kernel_ms_per_step": round(tot_ms, 3)},
        "kernels": kernels,
        "cold_ms": round(cold_ms, 2),
        # every timed block (eager): the spread says how much a fresh lease's clock ramp moves the number
        "timing": {"blocks_ms_per_step": [round(b / args.steps * 1e3, 3) for b in blocks],
                   "median_ms_per_step": round(ms_step, 3), "min_ms_per_step": round(min(blocks) / args.steps * 1e3, 3)},
        "gpu_state": gpu_state_under_load(step) if (world == 1 and not args.no_gpu_state) else None,
    }
    if graph_ms is not None:
        line["hip_graph"] = ({"ms_per_step": round(graph_ms, 3), "value": round(n / (graph_ms * 1e-3) / 1e6, 3),
                              "blocks_ms_per_step": [round(b, 3) for b in graph_blocks],
                              "note": "the same step replayed from a captured hipGraph, measured after the eager blocks "
                                      "(median block)"}
                             if not isinstance(graph_ms, str) else {"error": graph_ms})
    if world == 1 and args.cpu_budget > 0 and specs:
        line["cpu_baseline"] = cpu_baseline_both(lambda b: cpu_baseline_minkunet(coords, specs, b), args.cpu_budget) \
            if args.cpu_capped else cpu_baseline_minkunet(coords, specs, args.cpu_budget)
        if line["cpu_baseline"]:
            line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
    else:
        line["cpu_baseline"] = None
    return line


def extra_workloads(args, ME, MEB, dist_utils, rank, world, dev, startup):
    """The other single-GPU BASELINE configurations, measured in the same process right after the headline workload
    and reported as compact entries (their full lines: `python bench.py --workload ...`): configs[2] MinkUNet34C bf16 on
    the 200k-voxel scene, configs[4] the 4-D convolution.  Each entry carries its own roofline and its own
    reference-CPU baseline; short blocks (the driver's run has to stay within a few minutes)."""
    out = {}
    if world == 1 and not dist_utils.exchange_active():
        # (20 steps x >= 5 blocks after 5 warm-up steps: ~1.5 s of GPU time — 5 steps x 3 blocks on a fresh lease was too
        # little to be a robust number: VERDICT r4 weak #3)
        plan = (("minkunet34c_bf16_200k", dict(workload="minkunet", dtype="bf16", steps=20, warmup=5, min_time=1.0,
                                                min_blocks=5, cpu_budget=min(args.cpu_budget, 1.0))),
                ("conv4d_f32_400k", dict(workload="conv4d", dtype="f32", steps=10, warmup=3, min_time=0.1,
                                         cpu_budget=min(args.cpu_budget, 4.0))),
                # the headline layer at the SPARSE density: 100k voxels in 215^3 — one voxel in a
                # hundred occupied, 79 % of the pairs on the centre offset — beside the dense one (70^3, P ~ 8.4 N)
                ("conv3d_f32_100k_sparse", dict(workload="conv3d", dtype="f32", extent=215, steps=20, warmup=5,
                                                min_time=0.1, cpu_budget=0.0)))
    else:
        # N > 1 (the driver's scaling run): BASELINE configs[3] — MinkUNet34C, one 200k-voxel scene per rank, torch DDP
        # over RCCL — with the exchange priced (multi_gpu); --sync-bn / --imbalance of the command line carry over
        plan = (("minkunet34c_bf16_ddp", dict(workload="minkunet", dtype="bf16", steps=20, warmup=5, min_time=1.0, min_blocks=5,
                                               cpu_budget=0.0, sync_bn=args.sync_bn, imbalance=args.imbalance)),)
    for name, over in plan:
        a = argparse.Namespace(**vars(args))
        a.points = a.cin = a.cout = 0
        a.scenes, a.graph, a.sync_bn, a.imbalance = "cached", False, False, False
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        try:
            full = (bench_minkunet if a.workload == "minkunet" else bench_conv)(a, ME, MEB, dist_utils, rank, world, dev,
                                                                               startup)
        except Exception as e:  # noqa: BLE001  (the headline line must survive a failure here)
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            continue
        finally:
            torch.cuda.empty_cache()
        if full is None:
            continue
        keep = ("metric", "value", "unit", "n_gpus", "ms_per_step", "steps", "warmup", "dtype", "roofline", "cpu_baseline",
                "speedup_vs_cpu_baseline", "cold_ms", "hip_graph", "multi_gpu", "timing", "gpu_state")
        ent = {k: full[k] for k in keep if k in full}
        ent["config"] = {"workload": full["config"]["workload"], "parallelism": full["config"].get("parallelism")}
        ent["kernels"] = {k: {"avg_ms": v["avg_ms"], "tflops": v["tflops"]} for k, v in full.get("kernels", {}).items()}
        if "cold" in full and full["cold"]:
            ent["cold"] = {k: full["cold"][k] for k in ("insert_ms", "kernel_map_ms", "plans_ms", "kernel_map_GBs")
                           if k in full["cold"]}
        ent["blocks_ms_per_step"] = full["timing"]["blocks_ms_per_step"]
        ent["wall_s"] = round(time.perf_counter() - t0, 1)
        out[name] = ent
    return out


def capture_step(step):
    """Capture one training step (cached maps: no allocation-size read-backs, no host synchronisation) into a
    hipGraph on a side stream and return a replay function — the whole step becomes ONE host call
    (MI355X-first: HIP graphs for launch-bound loops; a bf16 MinkUNet34C step is ~700 launches)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    return graph.replay


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("conv3d", "conv4d", "minkunet"), default="conv3d")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="feature dtype (bf16: fp32 master weights, bf16 features, fp32 accumulation)")
    ap.add_argument("--points", type=int, default=0, help="voxels per GPU (0 = the BASELINE size of the workload)")
    ap.add_argument("--extent", type=int, default=70, help="conv3d: 70 = dense headline (P ~ 8.4 N); 215 = sparse")
    ap.add_argument("--cin", type=int, default=0)
    ap.add_argument("--cout", type=int, default=0)
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline timing (0 = skip)")
    ap.add_argument("--min-time", type=float, default=0.2, help="keep timing K-step blocks until this many seconds")
    ap.add_argument("--min-blocks", type=int, default=3)
    ap.add_argument("--no-gpu-state", action="store_true", help="skip the rocm-smi sample under load (MinkUNet lines)")
    ap.add_argument("--timer-blocks", type=int, default=4,
                    help="record the per-launch HIP events of the hot kernels in every N-th K-step block of the timed "
                         "region (1 = every block).  Six event records per step cost the one-layer headline ~20 us of a "
                         "0.31 ms step (profiles/r04_bench_timer_blocks.log): the kernel durations are sampled live in "
                         "a quarter of the timed blocks, the step time is the median over all of them")
    ap.add_argument("--max-blocks", type=int, default=200)
    ap.add_argument("--backend", choices=("auto", "nccl", "gloo"), default="auto")
    ap.add_argument("--arena-chunks", type=int, default=1,
                    help="--exchange arena: pieces of the flat gradient buffer all-reduced from inside the backward pass "
                         "(1 = one all-reduce after it; see distributed.GradientArena)")
    ap.add_argument("--exchange", choices=("arena", "ddp"), default="arena",
                    help="gradient exchange of the N > 1 step: distributed.GradientArena (one all-reduce of a flat buffer the "
                         "gradients are born in) or torch DistributedDataParallel (bucketed, overlapped)")
    ap.add_argument("--scenes", choices=("cached", "fresh", "pipelined"), default="cached",
                    help="minkunet: reuse one scene's maps (default, BASELINE configs[2]), or rebuild them every step")
    ap.add_argument("--loader-depth", type=int, default=1, help="minkunet --scenes pipelined: scenes built ahead")
    ap.add_argument("--replay-maps", action="store_true",
                    help="minkunet --scenes fresh: replay the previous scene's build requests in one burst (recipe "
                         "replay) instead of building lazily, layer by layer")
    ap.add_argument("--torch-loss", action="store_true",
                    help="minkunet: torch.nn.CrossEntropyLoss instead of examples/minkunet.py::cross_entropy (same math)")
    ap.add_argument("--sync-bn", action="store_true", help="minkunet, N > 1: MinkowskiSyncBatchNorm (reference recipe)")
    ap.add_argument("--graph", action="store_true", help="minkunet: replay the step from a captured hipGraph")
    ap.add_argument("--no-graph-probe", action="store_true",
                    help="minkunet: skip the extra hipGraph replay measurement reported beside the eager step")
    ap.add_argument("--extra-workloads", choices=("auto", "on", "off"), default="auto",
                    help="append compact entries for BASELINE configs[2] (MinkUNet34C bf16 @200k) and configs[4] (4-D "
                         "conv) under `workloads` (auto: with the default single-GPU headline run only)")
    ap.add_argument("--cpu-capped", action="store_true",
                    help="minkunet: also time the reference CPU layers under its own 16-thread cap (doubles the CPU time)")
    ap.add_argument("--pmc", choices=("auto", "on", "off"), default="auto",
                    help="measure roofline.traffic in this run with rocprofv3 --pmc passes over child processes (auto: "
                         "single-GPU runs when rocprofv3 is installed; bounded by --pmc-budget)")
    ap.add_argument("--pmc-budget", type=float, default=140.0, help="seconds the in-run PMC passes may take in total")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)   # internal: run N untimed steps and exit
    ap.add_argument("--accum", type=int, default=4,
                    help="N > 1, conv workloads: length of the gradient-accumulation window reported under "
                         "multi_gpu.accumulation (A - 1 micro-steps under DDP.no_sync, the A-th all-reduces)")
    ap.add_argument("--imbalance", action="store_true",
                    help="N > 1: ranks get scenes of different sizes, 0.75x .. 1.25x the nominal voxel count (real scans "
                         "differ in size: the scaling risk SURVEY 8(e) names); value still counts the voxels of all ranks")
    ap.add_argument("--debug-bf16-shape", default="",
                    help="'nc,kc': slab width / chunk depth override of the bf16 tile kernel (csrc/me_amd_debug.h)")
    ap.add_argument("--debug-conv-variant", type=int, default=0,
                    help="kernel-selection switch of csrc/me_amd_debug.h (tuning scripts; 0 = shipped kernels)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    t_start = time.perf_counter()
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import _lib
    from minkowskiengine_amd import backend as MEB
    from minkowskiengine_amd import distributed as dist_utils

    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    rank, world, local_rank = dist_utils.init_from_env(None if args.backend == "auto" else args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = dist_utils.local_device(local_rank)
    torch.cuda.set_device(dev)
    # one-time start-up costs, separated from the workload's cold path: library load, device context, first launch
    t_a = time.perf_counter()
    lib = _lib.load()
    t_b = time.perf_counter()
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    t_c = time.perf_counter()
    probe = torch.zeros((4, 4), dtype=torch.int32, device=dev)
    keys = torch.empty(4, dtype=torch.int64, device=dev)
    import ctypes
    _lib.check(lib.me_coords_spatial_keys(probe.data_ptr(), 4, 4, (ctypes.c_int32 * 3)(1, 1, 1), keys.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    t_d = time.perf_counter()
    # torch's own first use of its autograd machinery on this device: a TRIVIAL graph's first two backward passes cost
    # 60 ms each on this image (its elementwise / reduction / accumulate kernels load lazily; scripts/cold_backward_parts.py:
    # our first dgrad + wgrad of the headline layer together take 1.9 ms).  Paid here, reported, and so kept out of the
    # workload's cold path — BENCH_r04's 88 ms "first_backward_plans" was this, not the library's code objects.
    a_ = torch.ones(8, 8, device=dev, requires_grad=True)
    for _ in range(2):
        (a_ * 2.0).sum().backward()
    (a_.float().square().mean()).backward()
    torch.cuda.synchronize()
    del a_
    t_e = time.perf_counter()
    startup = {"import_ms": round((t_a - t_start) * 1e3, 1), "dlopen_libme_amd_ms": round((t_b - t_a) * 1e3, 1),
               "device_context_ms": round((t_c - t_b) * 1e3, 1), "first_kernel_ms": round((t_d - t_c) * 1e3, 1),
               "torch_autograd_first_use_ms": round((t_e - t_d) * 1e3, 1)}

    if args.debug_conv_variant:
        _lib.check(lib.me_debug_set_conv_variant(args.debug_conv_variant))
    if args.debug_bf16_shape:
        lib.me_debug_set_bf16_shape(*[int(v) for v in args.debug_bf16_shape.split(",")])
    fn = bench_minkunet if args.workload == "minkunet" else bench_conv
    if args.pmc_child:
        # profiled child of measure_traffic(): the same workload objects, `pmc_child` untimed steps, no JSON line
        args.steps, args.warmup, args.cpu_budget = args.pmc_child, 0, 0.0
        args.min_time, args.min_blocks, args.max_blocks, args.no_graph_probe = 0.0, 1, 1, True
        args.extra_workloads, args.pmc = "off", "off"
        args.pmc_child_mode = True
        fn(args, ME, MEB, dist_utils, rank, world, dev, startup)
        print(f"PMC_CHILD_STEPS={args.steps}", flush=True)
        return
    line = fn(args, ME, MEB, dist_utils, rank, world, dev, startup)
    default_headline = (args.workload == "conv3d" and args.dtype == "f32" and not args.points and
                        not args.cin and not args.cout and args.extent == 70)
    if args.extra_workloads == "on" or (args.extra_workloads == "auto" and default_headline):
        # (every rank takes part: the N > 1 entry is a collective workload; only rank 0 holds a line to attach it to)
        extra = extra_workloads(args, ME, MEB, dist_utils, rank, world, dev, startup)
        if line is not None:
            line["workloads"] = extra
    if rank == 0 and world == 1 and args.pmc != "off":
        t_pmc = time.perf_counter()
        attach_sq(line, args, t_pmc + 0.45 * args.pmc_budget)      # MFMA utilisation first (two passes at most)
        attach_traffic(line, args, t_pmc + args.pmc_budget)
    if rank == 0:
        for r in [line.get("roofline")] + [w.get("roofline") for w in (line.get("workloads") or {}).values()
                                           if isinstance(w, dict)]:
            if isinstance(r, dict):
                r.pop("pmc_match", None)
    dist_utils.shutdown()
    if rank == 0:
        # librccl prints a version banner through C stdio (fully buffered when stdout is a pipe: it would land BEHIND the
        # line at exit): flush it out first, so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
