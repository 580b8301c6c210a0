"""k_wgrad_bf16 against the wave-specialised k_wgrad_bf16_ws (me_debug_set_wgrad_ws 0 / 1 / 2): us per weight-gradient
launch (kernel + reduce) on the config-2 scene and the levels of the 200k-voxel MinkUNet scene (bf16).
usage: python scripts/wgrad_ws_sweep.py  (GPU, tuning build: scripts/build_debug.sh)"""
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
cases = [("config 2", make_scene(100000, 70, 0).to(dev), 1, 64, 128)]
coords = MU.synthetic_scene(200000, seed=0).to(dev)
mgr0 = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr0.insert_and_map(coords, [1, 1, 1], "")
levels = {1: coords}
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
for ts, cin, cout in ((1, 96, 96), (2, 96, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64), (4, 192, 128), (8, 128, 128), (8, 256, 256),
                      (8, 384, 256), (16, 256, 256), (16, 128, 256)):
    cases.append((f"unet ts{ts}", levels[ts], ts, cin, cout))
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
print(f"{'case':>10s} {'rows':>7s} {'layer':>10s} {'k_wgrad_bf16':>14s} {'ws (policy)':>14s} {'ws, 2 sets':>14s}   (us per launch incl. reduce)")
for name, c, ts, cin, cout in cases:
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
    gy = (torch.rand(c.shape[0], cout, generator=g) - 0.5).to(dev).bfloat16()
    w = (torch.rand(27, cin, cout, generator=g) - 0.5).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    k, _ = mgr.insert_and_map(c, [ts] * 3, "")
    km = mgr._kernel_map(k, k, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    cells = []
    for mode in (0, 1, 2):
        lib.me_debug_set_wgrad_ws(mode)
        cells.append(timed(lambda: MEB._conv_backward(x, gy, w, km, "mfma", need_grad_in=False)))
    print(f"{name:>10s} {c.shape[0]:7d} {str(cin) + '->' + str(cout):>10s} " + " ".join(f"{v:14.1f}" for v in cells), flush=True)
lib.me_debug_set_wgrad_ws(0)
