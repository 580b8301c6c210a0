"""Batches / groups per tile of the config-2 plan with row tiles and with spatial tiles (what the tile kernels pay
per tile is batches first, groups second)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from minkowskiengine_amd import backend as MEB
from bench import make_scene
dev = torch.device("cuda:0")
T = int(os.environ.get("TILE", "196"))
for order in ("rows", "spatial"):
    MEB._SPATIAL_MAPS, MEB._TILE_ORDER = True, order
    coords = make_scene(100000, 70, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    plan_src, plan_dst, batch_desc, tile_bptr, item_gptr = km.plan("out", T, 4)
    torch.cuda.synchronize()
    n_tiles = -(-100000 // T)
    bp = tile_bptr[:n_tiles + 1].cpu().numpy()
    nb = np.diff(bp)
    desc = batch_desc[:2 * bp[-1]].cpu().numpy().reshape(-1, 2)
    ng = desc[:, 1] & 255
    gpt = np.add.reduceat(ng, bp[:-1])
    hist = np.bincount(ng, minlength=5)
    print(f"{order:8s}: tiles {n_tiles}, batches total {bp[-1]} (per tile min {nb.min()} mean {nb.mean():.1f} max {nb.max()}), "
          f"groups total {ng.sum()} (per tile min {gpt.min()} mean {gpt.mean():.1f} max {gpt.max()}), "
          f"batches by groups 1..4: {hist[1:5].tolist()}, useful rows {km.n_pairs / (ng.sum() * 16):.3f}")
    # greedy heaviest-first makespan on 256 CUs (one tile per CU at a time), cost = batches / groups
    for name, cost in (("batches", nb.astype(float)), ("groups", gpt.astype(float)), ("2*batches+groups", 2.0 * nb + gpt)):
        load = np.zeros(256)
        for c in sorted(cost, reverse=True):
            load[np.argmin(load)] += c
        print(f"          LPT makespan by {name}: {load.max():.0f} (mean {cost.sum() / 256:.1f}, +{100 * (load.max() / (cost.sum() / 256) - 1):.1f} %)")
