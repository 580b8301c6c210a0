"""Ablation / tuning of the target-stationary convolution kernel on the config-2 workload:
variants (conv.hip VAR bits) x (tile height, batch capacity), forward and dgrad, HIP-event timed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene

dev = torch.device("cuda:0")
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
# "T:CAP" pairs; 0:0 = me_conv_plan_config
CONFIGS = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("CONFIGS", "0:0,196:6,196:5,196:4,131:4").split(",")]
EXTENTS = [int(v) for v in os.environ.get("EXTENTS", "70,215").split(",")]
CIN, COUT = int(os.environ.get("CIN", "64")), int(os.environ.get("COUT", "128"))
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


SPATIAL = [int(v) for v in os.environ.get("SPATIAL", "1").split(",")]
for extent, spatial in [(e, sp) for e in EXTENTS for sp in SPATIAL]:
    MEB._SPATIAL_TILES = bool(spatial)
    coords = make_scene(100000, extent, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    x = torch.rand(100000, CIN, device=dev)
    w = torch.rand(27, CIN, COUT, device=dev) - 0.5
    gy = torch.rand(100000, COUT, device=dev)
    flops = 2.0 * km.n_pairs * CIN * COUT
    print(f"== extent {extent} spatial tiles {spatial}: pairs {km.n_pairs}, auto (T, CAP) fwd {MEB.plan_config(100000, 27, km.n_pairs, CIN, COUT)}"
          f" dgrad {MEB.plan_config(100000, 27, km.n_pairs, COUT, CIN)}")
    ref = MEB._conv_forward(x, w, km, "naive")
    for var in VARIANTS:
        row = []
        for T, CAP in CONFIGS:
            MEB._TILE_ROWS, MEB._BATCH_GROUPS = T, CAP
            lib.me_debug_set_conv_variant(var)
            try:
                y = MEB._conv_forward(x, w, km, "mfma")
                err = float((y - ref).abs().max() / ref.abs().max())
                t = time_it(lambda: MEB._conv_forward(x, w, km, "mfma"))
                td = time_it(lambda: MEB._conv_target(gy, w, km, "in", km.n_in, transposed=True))
                row.append(f"T{T}c{CAP}: {t*1e3:.0f}us/{flops/t/1e9:.1f}TF d{td*1e3:.0f}us e{err:.0e}")
            except RuntimeError as ex:
                row.append(f"T{T}c{CAP}: ERR {str(ex)[-60:]}")
        print(f"var {var}: " + " | ".join(row), flush=True)
    lib.me_debug_set_conv_variant(0)
    MEB._TILE_ROWS = MEB._BATCH_GROUPS = 0
