"""bf16 weight gradient (k_wgrad_bf16 + k_wgrad_reduce) per MinkUNet34C level and channel shape by workgroups per CU the
pair ranges are sized for (me_debug_set_wgrad_config(0, wpc); 0 = shipped policy): fewer, longer ranges write fewer
partial images (one Cin x Cout fp32 image per range and offset it touches).  usage: python scripts/wgrad_ranges_sweep.py"""
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
k1, _ = mgr0.insert_and_map(coords, [1, 1, 1], "")
levels = {1: coords}
key = k1
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
ALL = {1: [(96, 96)], 2: [(96, 96), (32, 32)], 4: [(128, 128), (64, 64), (192, 128)],
       8: [(128, 128), (256, 256), (384, 256)], 16: [(256, 256), (128, 256)]}
WPC = [int(v) for v in os.environ.get("WPC", "0,1,2,3").split(",")]
REPS = int(os.environ.get("REPS", "20"))
print(f"{'level':>6s} {'rows':>7s} {'cin->cout':>10s} " + " ".join(f"{'wpc ' + str(w):>10s}" for w in WPC))
for ts in [int(l) for l in os.environ.get("LEVELS", "1,2,4,8,16").split(",")]:
    c = levels[ts]
    for cin, cout in ALL[ts]:
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
        w = (torch.rand(27, cin, cout, generator=torch.Generator().manual_seed(2)) - 0.5).to(dev)
        gy = (torch.rand(c.shape[0], cout, generator=g) - 0.5).to(dev).bfloat16()
        cells, ref = [], None
        for wpc in WPC:
            lib.me_debug_set_wgrad_config(0, wpc)
            mgr = MEB.CoordinateMapManagerGPU_c10()
            key, _ = mgr.insert_and_map(c, [ts] * 3, "")
            km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
            run = lambda: MEB._conv_backward(x, gy, w, km, "mfma", need_grad_in=False)[1]
            for _ in range(3):
                gw = run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(REPS):
                run()
            e.record()
            torch.cuda.synchronize()
            if ref is None:
                ref = gw.double()
                tag = ""
            else:
                tag = f" {float((gw.double() - ref).abs().max() / ref.abs().max()):.0e}"
            cells.append(f"{s.elapsed_time(e) / REPS * 1e3:6.1f}{tag}")
        print(f"{ts:6d} {c.shape[0]:7d} {str(cin) + '->' + str(cout):>10s} " + " ".join(f"{v:>10s}" for v in cells), flush=True)
lib.me_debug_set_wgrad_config(0, 0)
