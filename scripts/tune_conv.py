"""Ablation / tuning of the target-stationary convolution kernel on the config-2 workload:
variants (conv.hip VAR bits) x tile heights, forward kernel only, HIP-event timed."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minkowskiengine_amd as ME
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene

dev = torch.device("cuda:0")
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,16").split(",")]
TILES = [int(v) for v in os.environ.get("TILES", "64,96,112,128,131,160,196,256").split(",")]
lib = _lib.load()


def time_it(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for extent in (70, 215):
    coords = make_scene(100000, extent, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    x = torch.rand(100000, 64, device=dev)
    w = torch.rand(27, 64, 128, device=dev) - 0.5
    gy = torch.rand(100000, 128, device=dev)
    flops = 2.0 * km.n_pairs * 64 * 128
    print(f"== extent {extent}: pairs {km.n_pairs}, auto T fwd {lib.me_conv_choose_tile_rows(100000, 27, km.n_pairs, 64, 128)}"
          f" dgrad {lib.me_conv_choose_tile_rows(100000, 27, km.n_pairs, 128, 64)}")
    ref = None
    for var in VARIANTS:
        row = []
        for T in TILES:
            MEB._TILE_ROWS = T
            lib.me_debug_set_conv_variant(var)
            try:
                y = MEB._conv_forward(x, w, km, "mfma")
                if ref is None:
                    ref = y.clone()
                err = float((y - ref).abs().max())
                t = time_it(lambda: MEB._conv_forward(x, w, km, "mfma"))
                td = time_it(lambda: MEB._conv_target(gy, w, km, "in", km.n_in, transposed=True))
                row.append(f"T{T}: {t*1e3:.0f}us/{flops/t/1e9:.1f}TF d{td*1e3:.0f}us e{err:.0e}")
            except RuntimeError as ex:
                row.append(f"T{T}: ERR {str(ex)[:40]}")
        print(f"var {var}: " + " | ".join(row), flush=True)
    lib.me_debug_set_conv_variant(0)
    MEB._TILE_ROWS = 0
