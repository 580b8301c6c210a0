#!/usr/bin/env python
"""bench.py — the driver's benchmark contract for the MI355X sparse-convolution hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic input: forward + backward of ONE
MinkowskiConvolution (3-D, k = 3, s = 1, 64 -> 128 channels, fp32, no bias) on a 100k-voxel scene per
GPU (BASELINE.json configs[1]), kernel map cached (steady-state training layer), inputs resident in
HBM.  With N > 1 every rank owns its own scene (weak scaling, no data-path collective) and the
0.88 MB weight gradient is all-reduced over RCCL each step.  value = total voxels of all ranks /
max-over-ranks step time.

Other workloads (never the default; parity-test configurations timed for DESIGN.md):
    --workload conv4d     BASELINE configs[4]: 4-D k = 3 (K = 81), 400k voxels in 100^3 x 8 frames, 32 -> 64
    --workload minkunet   BASELINE configs[2]/[3]: MinkUNet34C forward + backward + SGD step on the 200k-voxel
                          plane-union scene of SURVEY 8(d); with N > 1 one scene per rank, all gradients
                          all-reduced in flat buckets over RCCL

Rank 0 prints ONE JSON line carrying `roofline` (dominant kernel: the target-stationary MFMA
convolution kernel, HIP-event timed inside the timed region) and `cpu_baseline` (the reference's own
CPU operators, oracle/_ref, timed on this box's host cores at N = 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0


def pmc_traffic(kernel, n, extent, cin, cout):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json; collected with scripts/gpu_pmc.sh on this exact workload), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        w = t["workload"]
        if (w["points"], w["extent"], w["cin"], w["cout"]) != (n, extent, cin, cout):
            return None
        k = t["kernels"][kernel]
        return int(k["fetch_bytes"] + k["write_bytes"])
    except Exception:  # noqa: BLE001
        return None


def cold_path(ME, MEB, feats, coords, dev, n, D=3, K=27, cin=64, cout=128):
    """Coordinate insertion + kernel-map build + tile plans of one scene, each timed with HIP events,
    with the achieved rate on SURVEY 8(d)'s algorithmic bytes (probes N*K x key bytes; insert N x key
    bytes + table)."""
    def timed(fn):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        torch.cuda.synchronize()
        return r, s.elapsed_time(e)

    out = {}
    best = {}
    for rep in range(3):   # first repetition includes allocator warm-up; report the best
        mgr = MEB.CoordinateMapManagerGPU_c10()
        (key, _), t_ins = timed(lambda: mgr.insert_and_map(coords, [1] * D, ""))
        km, t_km = timed(lambda: mgr._kernel_map(key, key, [3] * D, [1] * D, [1] * D, ME.RegionType.HYPER_CUBE,
                                                 None, False, False))
        def plans():
            for tgt, (cs, cd) in (("out", (cin, cout)), ("in", (cout, cin))):
                km.plan(tgt, *MEB.plan_config(n, K, km.n_pairs, cs, cd))
        _, t_plan = timed(plans)
        for name, t in (("insert_ms", t_ins), ("kernel_map_ms", t_km), ("plans_ms", t_plan)):
            best[name] = min(best.get(name, 1e9), t)
    key_bytes = 4 * (D + 1)
    out.update({k: round(v, 4) for k, v in best.items()})
    probe_bytes = n * K * key_bytes + 8 * km.n_pairs
    out["kernel_map_GBs"] = round(probe_bytes / (best["kernel_map_ms"] * 1e-3) / 1e9, 1)
    out["insert_GBs"] = round((n * key_bytes + 8 * 2 * n + 20 * n) / (best["insert_ms"] * 1e-3) / 1e9, 1)
    out["kernel_map_frac_of_hbm_peak"] = round(out["kernel_map_GBs"] / PEAK_HBM_GBS, 4)
    out["note"] = ("kernel_map_ms covers probe + scan + compaction incl. one host sync; bytes = N*K*4(D+1) probes + "
                   "8 B per pair written (SURVEY 8d); the map is built once per layer geometry and cached")
    return out


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


def kernel_table(timer, steps):
    """Per hot kernel: launches per step, mean ms per launch, algorithmic TFLOP/s over all its launches."""
    out = {}
    for name, (launches, total_ms, flops) in timer.totals().items():
        out[name] = {"launches_per_step": round(launches / steps, 2), "avg_ms": round(total_ms / launches, 4),
                     "ms_per_step": round(total_ms / steps, 4),
                     "tflops": round(flops / (total_ms * 1e-3) / 1e12, 2) if total_ms > 0 else None}
    return out


def run_timed(step, args, dist_utils, MEB, dev):
    for _ in range(args.warmup):
        step()
    timer = MEB.KernelTimer()
    dist_utils.barrier()
    torch.cuda.synchronize()
    MEB.KERNEL_TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist_utils.barrier()
    elapsed = time.perf_counter() - t0
    MEB.KERNEL_TIMER = None
    return dist_utils.max_over_ranks(elapsed, dev), timer


def bench_conv(args, ME, MEB, dist_utils, rank, world, dev):
    if args.workload == "conv4d":
        D, n, extent, cin, cout = 4, args.points or 400000, (100, 100, 100, 8), args.cin or 32, args.cout or 64
        cfg, ext_s = "BASELINE configs[4]", "[0,100)^3 x [0,8)"
    else:
        D, n, extent, cin, cout = 3, args.points or 100000, args.extent, args.cin or 64, args.cout or 128
        cfg, ext_s = "BASELINE configs[1]", f"[0,{args.extent})^3"
    K = 3 ** D
    coords = make_scene(n, extent, seed=rank, D=D)            # one independent scene per rank
    g = torch.Generator().manual_seed(1000 + rank)
    feats = torch.rand(n, cin, generator=g)
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, dimension=D, bias=False).to(dev)
    dist_utils.broadcast_parameters(conv)

    # cold path: coordinate insertion + kernel map + tile plans + first forward/backward
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    x = ME.SparseTensor(feats.to(dev).to(tdt), coords.to(dev), requires_grad=True)
    y = conv(x)
    y.F.sum().backward()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t0) * 1e3
    km = x.coordinate_manager._manager._kernel_map(x.coordinate_map_key, y.coordinate_map_key, [3] * D, [1] * D,
                                                   [1] * D, ME.RegionType.HYPER_CUBE, None, False, False)
    n_pairs = km.n_pairs
    grad_seed = torch.ones_like(y.F)

    def step():
        conv.kernel.grad = None
        x.F.grad = None
        out = conv(x)
        out.F.backward(grad_seed)
        dist_utils.allreduce_gradients(conv)

    elapsed, timer = run_timed(step, args, dist_utils, MEB, dev)
    total_points = dist_utils.sum_over_ranks(n, dev)
    pairs_all = dist_utils.sum_over_ranks(n_pairs, dev)
    cold = cold_path(ME, MEB, feats, coords.to(dev), dev, n, D, K, cin, cout) if rank == 0 else None
    if rank != 0:
        return None
    kernels = kernel_table(timer, args.steps)
    flops_per_launch = 2.0 * n_pairs * cin * cout     # each of forward / dgrad / wgrad
    achieved = kernels["conv_forward"]["tflops"]
    nc = 32 if 0 < cout % 64 <= 32 else 64                     # conv_variant() of csrc/conv.hip
    kc = 64 if cin % 64 == 0 else 32 if cin % 32 == 0 else 16
    bf16 = args.dtype == "bf16"
    if bf16:
        # conv_variant_bf16() of csrc/conv_bf16.hip
        kc = 128 if cin % 128 == 0 else 96 if cin % 96 == 0 else 32 if cin <= 32 else 64 if cin <= 64 else 128
    peak = PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_F32_MATRIX_TFLOPS
    esz = 2 if bf16 else 4
    line = {
        "metric": f"MinkowskiConvolution fwd+bwd Mpoints/sec ({n // 1000}k-pt {D}D, k=3)",
        "value": round(total_points / (elapsed / args.steps) / 1e6, 3),
        "unit": "Mpoints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"single MinkowskiConvolution {D}D k=3 s=1, {n} voxels/GPU uniform in "
                               f"{ext_s}, {cin}->{cout} ch, {'bf16 features / fp32 accumulate' if bf16 else 'fp32'}, "
                               f"kernel map cached ({cfg})",
                   "points_per_gpu": n, "pairs_per_gpu": n_pairs, "pairs_total": int(pairs_all),
                   "parallelism": f"scene-sharded dp{world}, RCCL all-reduce of the weight gradient"},
        "roofline": {"bound": "mfma", "kernel": f"k_conv_tile_{'bf16' if bf16 else 'f32'}<{nc},{kc}> (forward)",
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": pmc_traffic("k_conv_tile_bf16_forward" if bf16 else "k_conv_tile_f32", n, extent, cin,
                                            cout) if D == 3 else None,
                     "traffic_note": "HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE), rocprofv3 --pmc, "
                                     "profiles/pmc_traffic.json; compulsory bytes of the forward launch: "
                                     f"{int(esz * (n * cin + n * cout + K * cin * cout) + 8 * n_pairs)}",
                     "flops_per_launch": flops_per_launch},
        "kernels": kernels,
        "cold_ms": round(cold_ms, 2),
        "cold": cold,
    }
    if world == 1 and args.cpu_budget > 0:      # the reference CPU path is fp32 whatever our feature dtype
        line["cpu_baseline"] = cpu_baseline(coords, feats, conv.kernel.detach().float().cpu(), args.cpu_budget)
        line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
    else:
        line["cpu_baseline"] = None
    return line


def bench_minkunet(args, ME, MEB, dist_utils, rank, world, dev):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import minkunet as MU
    n = args.points or 200000
    coords = MU.synthetic_scene(n, seed=rank)
    n = coords.shape[0]
    g = torch.Generator().manual_seed(1000 + rank)
    feats = torch.rand(n, 3, generator=g)
    torch.manual_seed(0)
    net = MU.MinkUNet34C(3, 20, D=3).to(dev)
    dist_utils.broadcast_parameters(net)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    labels = torch.randint(0, 20, (n,), generator=g).to(dev)
    crit = torch.nn.CrossEntropyLoss()
    bf16 = args.dtype == "bf16"
    tdt = torch.bfloat16 if bf16 else torch.float32
    x = ME.SparseTensor(feats.to(dev).to(tdt), coords.to(dev))   # coordinate + kernel maps cached in x's manager

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x).F.float(), labels)
        loss.backward()
        dist_utils.allreduce_gradients(net)
        opt.step()

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t0) * 1e3
    elapsed, timer = run_timed(step, args, dist_utils, MEB, dev)
    total_points = dist_utils.sum_over_ranks(n, dev)
    if rank != 0:
        return None
    kernels = kernel_table(timer, args.steps)
    tot_ms = sum(k["ms_per_step"] for k in kernels.values())
    tot_flops = sum(f for _, _, f in timer.totals().values()) / args.steps
    achieved = round(tot_flops / (tot_ms * 1e-3) / 1e12, 2)
    return {
        "metric": "MinkUNet34C fwd+bwd+SGD Mpoints/sec (200k-pt synthetic scene)",
        "value": round(total_points / (elapsed / args.steps) / 1e6, 3),
        "unit": "Mpoints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"MinkUNet34C (3 -> 20 classes, {sum(p.numel() for p in net.parameters())} parameters) "
                               f"forward + cross-entropy + backward + SGD step, {n} voxels/GPU on a union of 9 planes "
                               f"in 400^3 (SURVEY 8d), {'bf16 activations / fp32 master weights and accumulation' if bf16 else 'fp32'}, "
                               "maps cached (BASELINE configs[2]; configs[3] with N = 8)",
                   "points_per_gpu": n,
                   "parallelism": f"scene-sharded dp{world}, RCCL all-reduce of all gradients in flat 25 MB buckets"},
        "roofline": {"bound": "mfma", "kernel": "all convolution launches of a step (k_conv_tile_f32 forward + dgrad, "
                                                "k_wgrad_f32)", "achieved": achieved,
                     "peak": PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / (PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_F32_MATRIX_TFLOPS), 4),
                     "traffic": None,
                     "flops_per_step": tot_flops, "conv_kernel_ms_per_step": round(tot_ms, 3)},
        "kernels": kernels,
        "cold_ms": round(cold_ms, 2),
        "cpu_baseline": None,
        "cpu_baseline_note": "the reference network needs the reference's Python package, which does not travel to "
                             "the GPU box; BASELINE.md §3: 21.7 s per iteration on the 8 survey-container cores",
    }


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
    args = ap.parse_args()

    import minkowskiengine_amd as ME
    from minkowskiengine_amd import backend as MEB
    from minkowskiengine_amd import distributed as dist_utils

    rank, world, local_rank = dist_utils.init_from_env()
    assert world == args.gpus or world == 1 and args.gpus == 1, \
        f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    fn = bench_minkunet if args.workload == "minkunet" else bench_conv
    line = fn(args, ME, MEB, dist_utils, rank, world, dev)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
