"""Split-K sweep of the bf16 tile kernel on the coarse MinkUNet34C levels of the 200k-voxel scene (tensor strides 8 and
16: 21k / 5k voxels): us per forward launch for G = 0 (unsplit) .. 8 offset groups, per channel shape.
usage: python scripts/splitk_sweep.py  (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ.setdefault("ME_AMD_HOST", "python")
import torch
from minkowskiengine_amd import backend as MEB, _lib
import minkunet as MU
lib = _lib.load()
dev = torch.device("cuda:0")
coords = MU.synthetic_scene(200000, seed=0).to(dev)
mgr0 = MEB.CoordinateMapManagerGPU_c10()
k1, _ = mgr0.insert_and_map(coords, [1, 1, 1], "")
levels = {}
key = k1
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
SHAPES = {16: [(256, 256), (128, 256), (256, 128)], 8: [(128, 128), (256, 256), (384, 256), (256, 384), (64, 128), (128, 64)],
          4: [(128, 128), (192, 128)]}
GS = [int(g) for g in os.environ.get("GS", "0,2,3,4,6,8").split(",")]
print(f"{'level':>6s} {'rows':>7s} {'cin->cout':>10s} " + " ".join(f"{'G=' + str(g):>16s}" for g in GS))
for ts, shapes in SHAPES.items():
    c = levels[ts]
    for cin, cout in shapes:
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
        w = (torch.rand(27, cin, cout, generator=g) - 0.5).to(dev)
        cells = []
        for G in GS:
            lib.me_debug_set_bf16_splitk(G)
            mgr = MEB.CoordinateMapManagerGPU_c10()
            key, _ = mgr.insert_and_map(c, [ts] * 3, "")
            km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
            T, _, sk = MEB.plan_config(km.n_out, km.volume, km.n_pairs, cin, cout, True, False, with_split_k=True)
            for _ in range(3):
                MEB._conv_forward(x, w, km, "mfma")
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                MEB._conv_forward(x, w, km, "mfma")
            e.record()
            torch.cuda.synchronize()
            cells.append(f"{s.elapsed_time(e) / 20 * 1e3:7.1f} T{T:<3d} g{sk}")
        print(f"{ts:6d} {c.shape[0]:7d} {str(cin) + '->' + str(cout):>10s} " + " ".join(f"{v:>16s}" for v in cells), flush=True)
lib.me_debug_set_bf16_splitk(-1)
