"""cProfile of MinkUNet34C forward passes with the maps rebuilt every step: where the host time of the cold path
goes (map building is host-bound: many small launches, allocations and host syncs)."""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
import minkowskiengine_amd.backend as MEB
import minkunet as MU
dev = torch.device("cuda:0")
coords = MU.synthetic_scene(200000, seed=0).to(dev)
feats = torch.rand(coords.shape[0], 3).to(dev).to(torch.bfloat16)
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
def fwd():
    with torch.no_grad():
        return net(ME.SparseTensor(feats, coords))
for _ in range(3): fwd()
torch.cuda.synchronize()
# wall time of the pieces, each synchronised
def timed(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
x = ME.SparseTensor(feats, coords)
with torch.no_grad(): net(x)
print(f"forward, maps cached : {timed(lambda: net(x) if torch.is_grad_enabled() else fwd_cached()) if False else 0:.2f}")
def fwd_cached():
    with torch.no_grad(): return net(x)
print(f"forward only, maps cached  {timed(fwd_cached):.2f} ms   maps rebuilt {timed(fwd):.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): fwd()
torch.cuda.synchronize(); pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28); print(s.getvalue()[:6000])
