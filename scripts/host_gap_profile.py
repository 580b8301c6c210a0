"""Host time of a config-2 layer step (forward + backward through the module API) with no device waits inside:
cProfile over STEPS steps, printed by internal time.  Used to compare tile orders (ME_AMD_SPATIAL_TILES,
ME_AMD_SPATIAL_MAPS / ME_AMD_TILE_ORDER): the kernels are the same, the step time is not."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minkowskiengine_amd as ME
from bench import make_scene

dev = torch.device("cuda:0")
extent = int(os.environ.get("EXTENT", "70"))
steps = int(os.environ.get("STEPS", "300"))
tdt = torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32
coords = make_scene(100000, extent, 0)
feats = torch.rand(100000, 64)
torch.manual_seed(0)
conv = ME.MinkowskiConvolution(64, 128, kernel_size=3, stride=1, dimension=3, bias=False).to(dev)
x = ME.SparseTensor(feats.to(dev).to(tdt), coords.to(dev), requires_grad=True)
y = conv(x)
grad_seed = torch.ones_like(y.F)


def step():
    conv.kernel.grad = None
    x.F.grad = None
    out = conv(x)
    out.F.backward(grad_seed)


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6 * (t1 - t0) / steps:.1f} us/step, with drain {1e6 * (t2 - t0) / steps:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get("TOP", "18")))
