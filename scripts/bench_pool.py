"""HBM-side rate of the pooling / broadcast / batch-norm / voxelisation kernels on BASELINE-sized inputs
(100k voxels in 70^3, 64 channels, fp32): algorithmic bytes (features read + written once, index
tables once) / HIP-event time, against the 8 TB/s HBM peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minkowskiengine_amd as ME
from bench import make_scene

dev = torch.device("cuda:0")
PEAK = 8000.0


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


n, c = 100000, 64
coords = make_scene(n, 70, 0).to(dev)
x = ME.SparseTensor(torch.rand(n, c, device=dev), coords)
rows = []


def report(name, nbytes, t):
    rows.append((name, nbytes / 1e6, t * 1e6, nbytes / t / 1e9, nbytes / t / 1e9 / PEAK))


with torch.no_grad():
    for name, mod, K in (("avg pool k3 s1", ME.MinkowskiAvgPooling(3, 1, dimension=3), 27),
                         ("max pool k3 s1", ME.MinkowskiMaxPooling(3, 1, dimension=3), 27),
                         ("sum pool k2 s2", ME.MinkowskiSumPooling(2, 2, dimension=3), 8)):
        y = mod(x)
        km = x.coordinate_manager._manager._kernel_map(x.coordinate_map_key, y.coordinate_map_key,
                                                         mod.kernel_generator.kernel_size, mod.kernel_generator.kernel_stride,
                                                         mod.kernel_generator.kernel_dilation, ME.RegionType.HYPER_CUBE, None,
                                                         False, True)
        t = timed(lambda: mod(x))
        # gathered feature rows (one per pair) + output rows + the neighbour table
        report(name, 4 * c * (km.n_pairs + y.F.shape[0]) + 4 * K * y.F.shape[0], t)
    gp = ME.MinkowskiGlobalAvgPooling()
    g = gp(x)
    report("global avg pool", 4 * c * n + 4 * n, timed(lambda: gp(x)))
    bc = ME.MinkowskiBroadcastAddition()
    report("broadcast add", 2 * 4 * c * n + 4 * n, timed(lambda: bc(x, g)))
    bn = ME.MinkowskiBatchNorm(c).to(dev)
    report("batch norm fwd (train)", 3 * 4 * c * n, timed(lambda: bn(x)))     # statistics pass + apply (read, write)
    dup = torch.cat([coords, coords[: n // 2]], 0)
    f = torch.rand(dup.shape[0], c, device=dev)
    t = timed(lambda: ME.SparseTensor(f, dup, quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE), iters=10)
    report("SparseTensor, 150k rows -> 100k voxels (insert + voxel average)", 16 * dup.shape[0] + 4 * c * (dup.shape[0] + n), t)
print(f"{'kernel path':66s} {'MB':>8s} {'us':>8s} {'GB/s':>8s} {'of HBM peak':>12s}")
for r in rows:
    print(f"{r[0]:66s} {r[1]:8.1f} {r[2]:8.1f} {r[3]:8.0f} {100 * r[4]:11.1f}%")
