"""256-channel layers of the coarse MinkUNet levels: k_conv_tile_bf16 with 256-channel chunks (deep pipeline, split-K by
policy) against the wave-specialised kernel on 128-channel chunks (me_debug_set_bf16_shape(128, 128)); us per forward /
input-gradient launch.  usage: python scripts/ws_kc_sweep.py  (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ.setdefault("ME_AMD_HOST", "python")
import torch
from minkowskiengine_amd import backend as MEB, _lib
import minkunet as MU
lib = _lib.load()
dev = torch.device("cuda:0")
coords = MU.synthetic_scene(200000, seed=0).to(dev)
mgr0 = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr0.insert_and_map(coords, [1, 1, 1], "")
levels = {}
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
def timed(fn, reps=30):
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
print(f"{'level':>6s} {'rows':>6s} {'layer':>9s}  {'kc 256 (policy)':>24s}  {'kc 128, ws':>24s}  {'kc 128, k_conv_tile_bf16':>24s}")
for ts, cin, cout in ((8, 256, 256), (16, 256, 256), (16, 256, 128), (8, 256, 128)):
    c = levels[ts]
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
    gy = (torch.rand(c.shape[0], cout, generator=g) - 0.5).to(dev).bfloat16()
    cells = []
    for shape, ws in (((0, 0), -1), ((128, 128), 1), ((128, 128), 0)):
        lib.me_debug_set_bf16_shape(*shape)
        lib.me_debug_set_bf16_ws(ws)
        w = (torch.rand(27, cin, cout, generator=g) - 0.5).to(dev)
        mgr = MEB.CoordinateMapManagerGPU_c10()
        k, _ = mgr.insert_and_map(c, [ts] * 3, "")
        km = mgr._kernel_map(k, k, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
        f = timed(lambda: MEB._conv_target(x, w, km, "out", km.n_out, name="conv_forward"))
        d = timed(lambda: MEB._conv_target(gy, w, km, "in", km.n_in, name="conv_dgrad", transposed=True))
        cf = MEB._conv_launch_cfg(km, "out", km.n_out, cin, cout, True)[1]
        cells.append(f"{f:6.1f} /{d:6.1f} T{cf[0]} g{cf[14]}")
    print(f"{ts:6d} {c.shape[0]:6d} {str(cin) + '->' + str(cout):>9s}  " + "  ".join(f"{v:>24s}" for v in cells), flush=True)
lib.me_debug_set_bf16_shape(0, 0)
lib.me_debug_set_bf16_ws(-1)
