import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
dev = torch.device("cuda:0")
lib = _lib.load()
MEB._WGRAD_TUNING = True
coords = make_scene(20000, 40, 0).to(dev)
mgr = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
for cin, cout in [(64, 128), (64, 256), (256, 256), (192, 128), (256, 64), (64, 192)]:
    x = torch.rand(20000, cin, device=dev).bfloat16()
    w = torch.rand(27, cin, cout, device=dev) - 0.5
    gy = (torch.rand(20000, cout, device=dev) - 0.5).bfloat16()
    print("shape", cin, cout, flush=True)
    lib.me_debug_set_wgrad_config(-1, 0)
    _, gw_ref = MEB._conv_backward(x, gy, w, km, "mfma")
    torch.cuda.synchronize()
    print("  old kernel ok", flush=True)
    for ks in [int(v) for v in os.environ.get("KS", "0").split(",")]:
        lib.me_debug_set_wgrad_config(ks, 0)
        print("  ksteps", ks, flush=True)
        _, gw = MEB._conv_backward(x, gy, w, km, "mfma")
        torch.cuda.synchronize()
        print("  new kernel ok, err", float((gw - gw_ref).abs().max() / gw_ref.abs().max()), flush=True)
