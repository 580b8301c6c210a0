"""Stacked-offset kernel (csrc/conv_stem.hip) against the tile-plan kernel on the MinkUNet stem and its relatives: forward
launch time (HIP events around 50 launches, plans / tables built before) with me_debug_set_stem(0 | 1), both hosts' shared
policy untouched otherwise.  Usage: python scripts/stem_sweep.py [n_voxels]"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import minkowskiengine_amd as ME            # noqa: E402
from minkowskiengine_amd import _lib        # noqa: E402
import minkunet as MU                       # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
coords = MU.synthetic_scene(n, seed=0).to(dev)
print(f"scene: {coords.shape[0]} voxels", flush=True)
SHAPES = [(3, 32, 5), (3, 32, 3), (3, 64, 3), (8, 64, 3), (4, 16, 3), (3, 64, 5)]
for cin, cout, ks in SHAPES:
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, dimension=3).to(dev)
    feats = torch.rand(coords.shape[0], cin, device=dev).to(torch.bfloat16)
    x = ME.SparseTensor(feats, coords)
    row = []
    outs = []
    for mode, g in ((0, 0), (1, 1), (1, 2), (1, 4)):
        lib.me_debug_set_stem(mode, g)
        with torch.no_grad():
            for _ in range(3):
                y = conv(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                y = conv(x)
            e1.record()
            torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 50 * 1e3)
        outs.append(y.F.float())
    err = max((outs[0] - o).abs().max().item() for o in outs[1:]) / max(1e-6, outs[0].abs().max().item())
    print(f"{cin:3d} -> {cout:3d}  k={ks}^3  layer (pad + launch) tile-plan {row[0]:8.1f} us   stacked g1 {row[1]:8.1f}  g2 {row[2]:8.1f}  "
          f"g4 {row[3]:8.1f} us   rel diff {err:.2e}", flush=True)
lib.me_debug_set_stem(-1, 0)
