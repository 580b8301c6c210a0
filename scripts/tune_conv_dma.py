"""Round 2: the LDS-DMA tile kernel (k_conv_tile_dma_f32, debug variants 3000 / 3064 / 3016) against the round-1
kernel (variant 0) on the config-2 workload and a few MinkUNet shapes: forward and dgrad, HIP-event timed, each
result checked against the atomics cross-check kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene

dev = torch.device("cuda:0")
lib = _lib.load()
# "variant:T:CAP" triples (T = CAP = 0: me_conv_plan_config)
SPECS = [tuple(int(v) for v in c.split(":")) for c in os.environ.get(
    "SPECS", "0:0:0,3000:196:4,3000:196:3,3000:98:3,3000:131:4,3064:196:3,3064:176:4,3064:131:4,3016:196:4").split(",")]
SHAPES = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("SHAPES", "70:64:128,215:64:128").split(",")]
N = int(os.environ.get("POINTS", "100000"))


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


for extent, cin, cout in SHAPES:
    coords = make_scene(N, extent, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    x = torch.rand(N, cin, device=dev)
    w = torch.rand(27, cin, cout, device=dev) - 0.5
    gy = torch.rand(N, cout, device=dev)
    flops = 2.0 * km.n_pairs * cin * cout
    ref = MEB._conv_forward(x, w, km, "naive")
    refd = torch.zeros(N, cin, device=dev)
    refd, _ = MEB._conv_backward(x, gy, w, km, "naive")
    print(f"== extent {extent} {cin}->{cout}: pairs {km.n_pairs}", flush=True)
    for var, T, CAP in SPECS:
        MEB._TILE_ROWS, MEB._BATCH_GROUPS = T, CAP
        lib.me_debug_set_conv_variant(var)
        try:
            y = MEB._conv_forward(x, w, km, "mfma")
            err = float((y - ref).abs().max() / ref.abs().max())
            t = time_it(lambda: MEB._conv_forward(x, w, km, "mfma"))
            msg = f"fwd {t*1e3:6.1f}us {flops/t/1e9:5.1f}TF err {err:.0e}"
        except RuntimeError as ex:
            msg = f"fwd ERR {str(ex)[-70:]}"
        try:
            gd = MEB._conv_target(gy, w, km, "in", km.n_in, transposed=True)
            errd = float((gd - refd).abs().max() / refd.abs().max())
            td = time_it(lambda: MEB._conv_target(gy, w, km, "in", km.n_in, transposed=True))
            msg += f" | dgrad {td*1e3:6.1f}us {flops/td/1e9:5.1f}TF err {errd:.0e}"
        except RuntimeError as ex:
            msg += f" | dgrad ERR {str(ex)[-70:]}"
        print(f"var {var:5d} T {T:3d} cap {CAP}: {msg}", flush=True)
    lib.me_debug_set_conv_variant(0)
    MEB._TILE_ROWS = MEB._BATCH_GROUPS = 0
