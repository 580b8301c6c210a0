"""Bank statistics of the tile plan: for every half group (8 slots) the multiplicity of the fullest
(target row mod 8) class = LDS passes of the 16-byte accumulate store."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene

dev = torch.device("cuda:0")
MEB._SPATIAL_TILES = os.environ.get("SPATIAL", "1") != "0"
coords = make_scene(100000, 70, 0).to(dev)
mgr = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
T, CAP = MEB.plan_config(100000, 27, km.n_pairs, 64, 128)
plan_src, plan_dst, batch_desc, tile_bptr, item_gptr = [t.cpu().numpy() for t in km.plan("out", T, CAP)]
ng = int(item_gptr[-1])
d = plan_dst[:ng * 16].reshape(ng, 2, 8)
valid = plan_src[:ng * 16].reshape(ng, 2, 8) >= 0
passes = np.zeros((ng, 2), np.int64)
for c in range(8):
    passes = np.maximum(passes, ((d % 8 == c) & valid).sum(-1))
print("T", T, "CAP", CAP, "groups", ng, "mean passes per half group", passes.mean(), "hist", np.bincount(passes.ravel()))
print("first groups:\n", d[:3], "\n", valid[:3])
