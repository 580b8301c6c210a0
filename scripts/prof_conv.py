"""Run only the hot kernels of the config-2 workload a few times (for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene

dev = torch.device("cuda:0")
lib = _lib.load()
var = int(os.environ.get("VARIANT", "0"))
T = int(os.environ.get("TILE", "0"))
extent = int(os.environ.get("EXTENT", "70"))
iters = int(os.environ.get("ITERS", "10"))
coords = make_scene(100000, extent, 0).to(dev)
mgr = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
tdt = torch.bfloat16 if os.environ.get("DTYPE", "f32") == "bf16" else torch.float32
x = torch.rand(100000, 64, device=dev).to(tdt)
w = torch.rand(27, 64, 128, device=dev) - 0.5
gy = torch.rand(100000, 128, device=dev).to(tdt)
MEB._TILE_ROWS = T
MEB._BATCH_GROUPS = int(os.environ.get("CAP", "0"))
lib.me_debug_set_conv_variant(var)
lib.me_debug_set_wgrad_order(int(os.environ.get("WGRAD_ORDER", "0")))
for _ in range(iters):
    y = MEB._conv_forward(x, w, km, "mfma")
    if os.environ.get("BWD", "1") == "1":
        MEB._conv_backward(x, gy, w, km, "mfma")
torch.cuda.synchronize()
print("done", km.n_pairs)
