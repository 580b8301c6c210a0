"""One convolution layer on a level of the MinkUNet34C scene (for rocprofv3 --pmc passes): LEVEL = tensor stride of the
map (1, 2, 4, 8, 16), CIN / COUT, DTYPE = bf16 | f32; forward, input gradient and weight gradient ITERS times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ.setdefault("ME_AMD_HOST", "python")
import torch
from minkowskiengine_amd import backend as MEB
import minkunet as MU

dev = torch.device("cuda:0")
level, cin, cout = int(os.environ.get("LEVEL", "8")), int(os.environ.get("CIN", "256")), int(os.environ.get("COUT", "256"))
iters = int(os.environ.get("ITERS", "5"))
tdt = torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32
coords = MU.synthetic_scene(200000, seed=0).to(dev)
mgr = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
ts = 1
while ts < level:
    key = mgr.stride(key, [2, 2, 2])
    ts *= 2
km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
n = km.n_in
x = (torch.rand(n, cin, device=dev) - 0.5).to(tdt)
w = torch.rand(27, cin, cout, device=dev) - 0.5
gy = (torch.rand(n, cout, device=dev) - 0.5).to(tdt)
for _ in range(iters):
    MEB._conv_forward(x, w, km, "mfma")
    MEB._conv_backward(x, gy, w, km, "mfma")
torch.cuda.synchronize()
print("done", n, km.n_pairs)
