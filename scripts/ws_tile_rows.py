"""Tile height of the wave-specialised bf16 kernel: us per forward launch by forced tile_rows (ME_AMD_TILE_ROWS) on a few
MinkUNet layers; 0 = the plan policy.  usage: python scripts/ws_tile_rows.py  (GPU)"""
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
coords = MU.synthetic_scene(200000, seed=0).to(dev)
mgr0 = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr0.insert_and_map(coords, [1, 1, 1], "")
levels = {1: coords}
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
cases = [("config 2", make_scene(100000, 70, 0).to(dev), 1, 64, 128), ("unet ts2", levels[2], 2, 96, 96), ("unet ts2", levels[2], 2, 128, 96),
         ("unet ts4", levels[4], 4, 128, 128), ("unet ts4", levels[4], 4, 64, 64), ("unet ts4", levels[4], 4, 192, 128),
         ("unet ts8", levels[8], 8, 128, 128), ("unet ts8", levels[8], 8, 384, 256)]
TS = (0, 64, 96, 112, 128, 144, 160, 176, 192, 224, 256)
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print(f"{'case':>10s} {'layer':>10s} " + " ".join(f"{('T ' + str(t)) if t else 'policy':>8s}" for t in TS))
for name, c, ts, cin, cout in cases:
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
    w = (torch.rand(27, cin, cout, generator=g) - 0.5).to(dev)
    cells = []
    for T in TS:
        MEB._TILE_ROWS = T
        mgr = MEB.CoordinateMapManagerGPU_c10()
        k, _ = mgr.insert_and_map(c, [ts] * 3, "")
        km = mgr._kernel_map(k, k, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
        try:
            f = timed(lambda: MEB._conv_target(x, w, km, "out", km.n_out, name="conv_forward"))
            tf = MEB._conv_launch_cfg(km, "out", km.n_out, cin, cout, True)[1][0]
            cells.append(f"{f:5.1f}" + (f"@{tf}" if T == 0 else ""))
        except Exception as e:
            cells.append("err")
    print(f"{name:>10s} {str(cin) + '->' + str(cout):>10s} " + " ".join(f"{v:>8s}" for v in cells), flush=True)
