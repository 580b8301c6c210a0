"""fp32 weight gradient of the config-2 layer (and two MinkUNet shapes): XCD-aware range order against launch order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
dev = torch.device("cuda:0")
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
    return s.elapsed_time(e) / iters * 1e3


for extent, cin, cout in ((70, 64, 128), (70, 128, 128), (70, 32, 64), (215, 64, 128)):
    coords = make_scene(100000, extent, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    x = torch.rand(100000, cin, device=dev)
    gy = torch.rand(100000, cout, device=dev)
    w = torch.rand(27, cin, cout, device=dev) - 0.5
    res, outs = {}, {}
    for mode in (-1, 0, -1, 0):
        lib.me_debug_set_wgrad_order(mode)
        MEB.KERNEL_TIMER = MEB.KernelTimer()
        for _ in range(12):
            _, gw = MEB._conv_backward(x, gy, w, km, "mfma")
        torch.cuda.synchronize()
        t = MEB.KERNEL_TIMER.summary()["conv_wgrad"][1] * 1e3
        MEB.KERNEL_TIMER = None
        res.setdefault(mode, []).append(t)
        outs[mode] = gw
    lib.me_debug_set_wgrad_order(0)
    same = torch.equal(outs[-1], outs[0])
    print(f"extent {extent} {cin}->{cout}: launch order {min(res[-1]):.1f} us, XCD-aware {min(res[0]):.1f} us "
          f"(runs {res}), bit-identical {same}", flush=True)
