"""k_wgrad_f32x3: workgroups per CU (ranges = CUs x wpc) on the config-2 layer and two other shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
dev = torch.device("cuda:0")
lib = _lib.load()
MEB._WGRAD_TUNING = True
SHAPES = [tuple(int(v) for v in t.split(':')) for t in os.environ.get('SHAPES', '70:64:128,70:128:128,70:64:64,215:64:128').split(',')]
WPCS = [int(v) for v in os.environ.get('WPCS', '2,3,4,6,9').split(',')]
for extent, cin, cout in SHAPES:
    coords = make_scene(100000, extent, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    x = torch.rand(100000, cin, device=dev)
    gy = torch.rand(100000, cout, device=dev)
    w = torch.rand(27, cin, cout, device=dev) - 0.5
    line = []
    for depth, wpc in [(-3, 0)] + [(-4, w) for w in WPCS]:
        lib.me_debug_set_wgrad_config(depth, wpc)
        km._launch_cache.clear()
        best = 1e9
        for rep in range(2):
            MEB.KERNEL_TIMER = MEB.KernelTimer()
            for _ in range(10):
                MEB._conv_backward(x, gy, w, km, "mfma")
            torch.cuda.synchronize()
            best = min(best, MEB.KERNEL_TIMER.summary()["conv_wgrad"][1] * 1e3)
            MEB.KERNEL_TIMER = None
        line.append(f"{'mfma' if depth == -3 else 'x3 wpc ' + str(wpc)}: {best:.1f}")
    lib.me_debug_set_wgrad_config(0, 0)
    print(f"extent {extent} {cin}->{cout} (wgrad + reduce, us): " + ", ".join(line), flush=True)
