"""A/B of the two bf16 weight-gradient kernels (k_wgrad_bf16: bf16 MFMA through LDS transposing reads;
k_wgrad_f32<__bf16>: fp32 MFMA fed with converted rows), checked against the fp32 kernel on the rounded
inputs.  CONFIGS = depth:wgs_per_cu list (depth -1 selects the fp32-MFMA kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene

dev = torch.device("cuda:0")
lib = _lib.load()
MEB._WGRAD_TUNING = True
shapes = [(64, 128), (128, 64), (32, 32), (32, 64), (96, 96), (256, 256), (64, 64)]
CONFIGS = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("CONFIGS", "0:0,0:3,-1:0").split(",")]
for extent in (70, 215):
    coords = make_scene(100000, extent, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    for cin, cout in shapes:
        x = torch.rand(100000, cin, device=dev).bfloat16()
        w = torch.rand(27, cin, cout, device=dev) - 0.5
        gy = (torch.rand(100000, cout, device=dev) - 0.5).bfloat16()
        flops = 2.0 * km.n_pairs * cin * cout
        _, gw_ref = MEB._conv_backward(x.float(), gy.float(), w, km, "mfma")
        res = []
        for depth, wpc in CONFIGS:
            lib.me_debug_set_wgrad_config(depth, wpc)
            _, gw = MEB._conv_backward(x, gy, w, km, "mfma")
            _, gw2 = MEB._conv_backward(x, gy, w, km, "mfma")
            err = float((gw - gw_ref).abs().max() / gw_ref.abs().max())
            rep = bool((gw == gw2).all())
            MEB.KERNEL_TIMER = MEB.KernelTimer()
            for _ in range(20):
                MEB._conv_backward(x, gy, w, km, "mfma")
            torch.cuda.synchronize()
            t = MEB.KERNEL_TIMER.summary()["conv_wgrad"][1]
            MEB.KERNEL_TIMER = None
            res.append(f"d{depth}w{wpc}: {t*1e3:.0f}us/{flops/t/1e9:.0f}TF err {err:.0e}{'' if rep else ' NONREPRO'}")
        lib.me_debug_set_wgrad_config(0, 0)
        print(f"extent {extent} {cin}->{cout}: " + " | ".join(res), flush=True)
