"""Dispatch order of the tiles (me_debug_set_tile_dispatch: 0 = heaviest first over the launch, 1 = contiguous tile
chunks per XCD) x tile order (rows / spatial): us per forward and per input-gradient launch on the config-2 scene
(fp32 and bf16, 64 -> 128) and on the levels of the 200k-voxel MinkUNet scene (bf16).
usage: python scripts/tile_dispatch_sweep.py  (GPU)"""
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

cases = [("config 2", make_scene(100000, 70, 0).to(dev), 1, 64, 128, torch.float32),
         ("config 2", make_scene(100000, 70, 0).to(dev), 1, 64, 128, torch.bfloat16)]
coords = MU.synthetic_scene(200000, seed=0).to(dev)
mgr0 = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr0.insert_and_map(coords, [1, 1, 1], "")
levels = {1: coords}
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
for ts, cin, cout in ((1, 96, 96), (2, 96, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64), (8, 128, 128), (8, 256, 256), (16, 256, 256)):
    cases.append((f"unet ts{ts}", levels[ts], ts, cin, cout, torch.bfloat16))

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

MODES = [("auto", 0), ("auto", 1), ("spatial", 0), ("spatial", 1)]
print(f"{'case':>10s} {'rows':>7s} {'layer':>14s}  " + "  ".join(f"{o + '/' + ('xcd' if d else 'lpt'):>15s}" for o, d in MODES) + "   (forward / dgrad us)")
for name, c, ts, cin, cout, dt in cases:
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).to(dt)
    gy = (torch.rand(c.shape[0], cout, generator=g) - 0.5).to(dev).to(dt)
    w = (torch.rand(27, cin, cout, generator=g) - 0.5).to(dev)
    cells = []
    for order, disp in MODES:
        MEB._TILE_ORDER = order
        lib.me_debug_set_tile_dispatch(disp)
        mgr = MEB.CoordinateMapManagerGPU_c10()
        k, _ = mgr.insert_and_map(c, [ts] * 3, "")
        km = mgr._kernel_map(k, k, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
        f = timed(lambda: MEB._conv_target(x, w, km, "out", km.n_out, name="conv_forward"))
        d = timed(lambda: MEB._conv_target(gy, w, km, "in", km.n_in, name="conv_dgrad", transposed=True))
        cells.append(f"{f:6.1f} /{d:6.1f}")
    print(f"{name:>10s} {c.shape[0]:7d} {str(cin) + '->' + str(cout) + (' f32' if dt == torch.float32 else ' bf16'):>14s}  " + "  ".join(f"{v:>15s}" for v in cells), flush=True)
lib.me_debug_set_tile_dispatch(0)
