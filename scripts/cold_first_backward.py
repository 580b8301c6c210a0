"""Where the first backward pass of a process spends its time (BENCH cold_breakdown_ms.first_backward_plans = 86 ms): wall
time of import, me_preload, first forward, first backward and its parts, with ME_AMD_PRELOAD=0/1 from the environment."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import torch
torch.zeros(1, device="cuda").item()
t1 = time.perf_counter()
import minkowskiengine_amd as ME
from minkowskiengine_amd import _lib
lib = _lib.load()
t2 = time.perf_counter()
sys.path.insert(0, ROOT)
from bench import make_scene
dev = torch.device("cuda:0")
coords = make_scene(100000, 70, 0).to(dev)
feats = torch.rand(100000, 64, device=dev)
conv = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).to(dev)
torch.cuda.synchronize(); t3 = time.perf_counter()
x = ME.SparseTensor(feats, coords, requires_grad=True)
y = conv(x)
torch.cuda.synchronize(); t4 = time.perf_counter()
g = torch.ones_like(y.F)
torch.cuda.synchronize(); t5 = time.perf_counter()
y.F.backward(g)
torch.cuda.synchronize(); t6 = time.perf_counter()
y = conv(x); torch.cuda.synchronize(); t7 = time.perf_counter()
y.F.backward(g); torch.cuda.synchronize(); t8 = time.perf_counter()
print(f"preload={os.environ.get('ME_AMD_PRELOAD', '1')} host={ME.get_host()}: torch+context {1e3 * (t1 - t0):.0f} ms, import+load(+preload) {1e3 * (t2 - t1):.0f}, "
      f"first forward {1e3 * (t4 - t3):.1f}, first backward {1e3 * (t6 - t5):.1f}, second forward {1e3 * (t7 - t6):.2f}, second backward {1e3 * (t8 - t7):.2f}")
