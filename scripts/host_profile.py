"""Host-side (Python / ctypes / torch dispatch) cost of one conv forward+backward step: cProfile of 300 steps."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minkowskiengine_amd as ME
from bench import make_scene
dev = torch.device("cuda:0")
dt = torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32
coords = make_scene(100000, 70, 0).to(dev)
x = ME.SparseTensor(torch.rand(100000, 64, device=dev).to(dt), coords, requires_grad=True)
conv = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).to(dev)
g = torch.ones(100000, 128, device=dev, dtype=dt)
def step():
    conv.kernel.grad = None
    x.F.grad = None
    conv(x).F.backward(g)
for _ in range(20): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300): step()
t1 = time.perf_counter()          # host time to ENQUEUE 300 steps (no sync)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6*(t1-t0)/300:.0f} us/step, with drain {1e6*(t2-t0)/300:.0f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
