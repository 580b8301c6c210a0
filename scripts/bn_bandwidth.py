"""Batch-norm kernels against the HBM roof: HIP-event time per API call (stats = partial + final, apply, backward =
partial + final + apply) on the (rows, channels) shapes of a MinkUNet34C step at 200k voxels, bf16.  Algorithmic bytes:
stats n*c*2, apply 2*n*c*2 (3x with the residual branch), backward 5*n*c*2 (partial reads x, dy; apply reads x, dy,
writes dx)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB
dev = torch.device("cuda:0")
shapes = [(200000, 32), (200000, 96), (160907, 96), (160907, 32), (79572, 64), (79572, 128), (21176, 128), (21176, 256), (4977, 256)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print(f"{'rows':>8s} {'c':>4s} {'MB':>6s} | {'stats us':>8s} {'GB/s':>6s} | {'apply us':>8s} {'GB/s':>6s} | {'apply+res':>9s} {'GB/s':>6s} | {'bwd us':>8s} {'GB/s':>6s} | {'bwd+res':>8s} {'GB/s':>6s}")
for n, c in shapes:
    x = (torch.randn(n, c, device=dev)).bfloat16(); dy = torch.randn(n, c, device=dev).bfloat16(); sk = torch.randn(n, c, device=dev).bfloat16()
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    mean, rstd = MEB.bn_stats(x, 1e-5, 0.1)
    y = MEB.bn_apply_residual(x, sk, mean, rstd, g, b, True)
    mb = n * c * 2 / 1e6
    t1 = timed(lambda: MEB.bn_stats(x, 1e-5, 0.1))
    t2 = timed(lambda: MEB.bn_apply(x, mean, rstd, g, b, True))
    t3 = timed(lambda: MEB.bn_apply_residual(x, sk, mean, rstd, g, b, True))
    t4 = timed(lambda: MEB.bn_backward(x, dy, mean, rstd, g, b, True))
    t5 = timed(lambda: MEB.bn_backward_residual(x, dy, y, mean, rstd, g, b, True, True))
    print(f"{n:8d} {c:4d} {mb:6.1f} | {t1:8.1f} {mb/t1*1e3:6.0f} | {t2:8.1f} {2*mb/t2*1e3:6.0f} | {t3:9.1f} {3*mb/t3*1e3:6.0f} | {t4:8.1f} {5*mb/t4*1e3:6.0f} | {t5:8.1f} {8*mb/t5*1e3:6.0f}")
