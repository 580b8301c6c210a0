"""The cold path of config 2 (insert + kernel map + forward / dgrad plans of a 100k-voxel scene), five times, for a
rocprofv3 --kernel-trace --stats run: which kernels the 0.1 + 0.08 + 0.29 ms are made of."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import minkowskiengine_amd as ME
from minkowskiengine_amd import backend as MEB
dev = torch.device("cuda:0")
n = int(os.environ.get("N", 100000)); D = int(os.environ.get("D", 3)); ext = int(os.environ.get("EXTENT", 70))
coords = bench.make_scene(n, ext, 0, D=D).to(dev)
feats = torch.rand(coords.shape[0], 64, device=dev)
for rep in range(5):
    r = bench.cold_path(ME, MEB, feats, coords, dev, coords.shape[0], D=D, K=3 ** D)
    torch.cuda.synchronize()
print({k: v for k, v in r.items() if k != "note"})
