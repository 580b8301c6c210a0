"""Time / check the wgrad kernel on the config-2 workload (HIP-event timed, checked against the
VALU cross-check kernel)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene

dev = torch.device("cuda:0")
lib = _lib.load()
MEB._WGRAD_TUNING = True


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


shapes = [(64, 128), (128, 64), (32, 32), (96, 96), (3, 32), (256, 256)]
CONFIGS = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("CONFIGS", "0:0").split(",")]
for extent in (70, 215):
    coords = make_scene(100000, extent, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    for cin, cout in shapes:
        x = torch.rand(100000, cin, device=dev)
        w = torch.rand(27, cin, cout, device=dev) - 0.5
        gy = torch.rand(100000, cout, device=dev) - 0.5
        flops = 2.0 * km.n_pairs * cin * cout
        _, gw = MEB._conv_backward(x, gy, w, km, "mfma")
        _, gw_ref = MEB._conv_backward(x, gy, w, km, "naive")
        err = float((gw - gw_ref).abs().max() / gw_ref.abs().max())
        _, gw2 = MEB._conv_backward(x, gy, w, km, "mfma")
        rep = bool((gw == gw2).all())
        res = []
        for depth, wpc in CONFIGS:
            lib.me_debug_set_wgrad_config(depth, wpc)
            for _ in range(3):
                MEB._conv_backward(x, gy, w, km, "mfma")
            MEB.KERNEL_TIMER = MEB.KernelTimer()
            for _ in range(20):
                MEB._conv_backward(x, gy, w, km, "mfma")
            torch.cuda.synchronize()
            tm = MEB.KERNEL_TIMER.summary()
            MEB.KERNEL_TIMER = None
            t = tm["conv_wgrad"][1]
            res.append(f"d{depth}w{wpc}: {t*1e3:.0f}us/{flops/t/1e9:.1f}TF")
        lib.me_debug_set_wgrad_config(0, 0)
        print(f"extent {extent} {cin}->{cout}: relerr {err:.1e} repro {rep} | " + " | ".join(res), flush=True)
