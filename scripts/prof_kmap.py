"""Build the coordinate map + kernel map (+ plans) of a workload a few times, for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB
from bench import make_scene

dev = torch.device("cuda:0")
wl = os.environ.get("WORKLOAD", "conv3d")
if wl == "conv4d":
    D, n, extent = 4, 400000, (100, 100, 100, 8)
else:
    D, n, extent = 3, 100000, int(os.environ.get("EXTENT", "70"))
coords = make_scene(n, extent, 0, D=D).to(dev)
for rep in range(int(os.environ.get("REPS", "5"))):
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1] * D, "")
    km = mgr._kernel_map(key, key, [3] * D, [1] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
    if os.environ.get("PLANS", "1") == "1":
        for tgt in ("out", "in"):
            km.plan(tgt, *MEB.plan_config(n, 3 ** D, km.n_pairs, 64, 128))
torch.cuda.synchronize()
print("pairs", km.n_pairs)
