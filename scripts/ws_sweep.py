"""k_conv_tile_bf16 against the wave-specialised k_conv_tile_bf16_ws (me_debug_set_bf16_ws 0 / 1): us per forward and
input-gradient launch on the config-2 scene and on the levels of the 200k-voxel MinkUNet scene (bf16).
usage: python scripts/ws_sweep.py  (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ.setdefault("ME_AMD_HOST", "python")
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
import minkunet as MU
lib = _lib.load()
dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "20"))
cases = [("config 2", make_scene(100000, 70, 0).to(dev), 1, 64, 128)]
coords = MU.synthetic_scene(200000, seed=0).to(dev)
mgr0 = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr0.insert_and_map(coords, [1, 1, 1], "")
levels = {1: coords}
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
for ts, cin, cout in ((1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 128, 96), (4, 128, 128), (4, 64, 64), (4, 192, 128), (8, 128, 128),
                      (8, 384, 256), (8, 256, 256), (16, 128, 256), (16, 256, 256)):
    cases.append((f"unet ts{ts}", levels[ts], ts, cin, cout))

def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS * 1e3

print(f"{'case':>10s} {'rows':>7s} {'layer':>10s}  {'k_conv_tile_bf16':>22s}  {'ws, 4 multipliers':>22s}  {'ws, 8 on 128 columns':>22s}   (forward / dgrad us, tile rows)")
for name, c, ts, cin, cout in cases:
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
    gy = (torch.rand(c.shape[0], cout, generator=g) - 0.5).to(dev).bfloat16()
    w = (torch.rand(27, cin, cout, generator=g) - 0.5).to(dev)
    cells = []
    for mode, depth, ncw in ((0, 4, 4), (1, 4, 4), (1, 4, 8)):
        lib.me_debug_set_bf16_ws(mode)
        lib.me_debug_set_bf16_ws_depth(depth)
        lib.me_debug_set_bf16_ws_ncw(ncw)
        mgr = MEB.CoordinateMapManagerGPU_c10()
        k, _ = mgr.insert_and_map(c, [ts] * 3, "")
        km = mgr._kernel_map(k, k, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
        f = timed(lambda: MEB._conv_target(x, w, km, "out", km.n_out, name="conv_forward"))
        d = timed(lambda: MEB._conv_target(gy, w, km, "in", km.n_in, name="conv_dgrad", transposed=True))
        tf = MEB._conv_launch_cfg(km, "out", km.n_out, cin, cout, True)[1][0]
        td = MEB._conv_launch_cfg(km, "in", km.n_in, cout, cin, True)[1][0]
        cells.append(f"{f:6.1f} /{d:6.1f} T{tf}/{td}")
    print(f"{name:>10s} {c.shape[0]:7d} {str(cin) + '->' + str(cout):>10s}  " + "  ".join(f"{v:>22s}" for v in cells), flush=True)
lib.me_debug_set_bf16_ws(-1)
lib.me_debug_set_bf16_ws_ncw(0)
