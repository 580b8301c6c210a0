"""Split-K / shape sweep of the bf16 tile kernel on the coarse MinkUNet34C levels of the 200k-voxel scene (tensor strides
8 and 16: 21k / 5k voxels): us per forward launch per configuration.  A configuration is "nc,kc,G,mode" — slab width and
chunk depth overrides (0 = policy), offset groups (0 = unsplit, -1 = policy), mode 1 = keep the unsplit tile height.
usage: CONFIGS="0,0,0,0;0,0,2,0;128,128,2,1" python scripts/splitk_sweep.py  (GPU)"""
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
ALL = {16: [(256, 256), (128, 256), (256, 128)], 8: [(128, 128), (256, 256), (384, 256), (256, 384)], 4: [(128, 128)]}
want = os.environ.get("LEVELS", "16,8")
SHAPES = {int(l): ALL[int(l)] for l in want.split(",")}
CONFIGS = [tuple(int(v) for v in c.split(",")) for c in
           os.environ.get("CONFIGS", "0,0,0,0;0,0,2,0;0,0,4,0;128,128,0,0;128,128,2,1;128,128,3,1;128,128,2,0").split(";")]
REPS = int(os.environ.get("REPS", "20"))
print(f"{'level':>6s} {'rows':>7s} {'cin->cout':>10s} " + " ".join(f"{'/'.join(str(v) for v in c):>18s}" for c in CONFIGS))
for ts, shapes in SHAPES.items():
    c = levels[ts]
    for cin, cout in shapes:
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
        cells = []
        for nc, kc, G, mode in CONFIGS:
            lib.me_debug_set_bf16_shape(nc, kc)
            lib.me_debug_set_bf16_splitk(G)
            lib.me_debug_set_bf16_splitk_mode(mode)
            w = (torch.rand(27, cin, cout, generator=g) - 0.5).to(dev)      # (packed images depend on the shape)
            mgr = MEB.CoordinateMapManagerGPU_c10()
            key, _ = mgr.insert_and_map(c, [ts] * 3, "")
            km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
            T, _, sk = MEB.plan_config(km.n_out, km.volume, km.n_pairs, cin, cout, True, False, with_split_k=True)
            for _ in range(3):
                MEB._conv_forward(x, w, km, "mfma")
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(REPS):
                MEB._conv_forward(x, w, km, "mfma")
            e.record()
            torch.cuda.synchronize()
            cells.append(f"{s.elapsed_time(e) / REPS * 1e3:7.1f} T{T:<3d} g{sk}")
        print(f"{ts:6d} {c.shape[0]:7d} {str(cin) + '->' + str(cout):>10s} " + " ".join(f"{v:>18s}" for v in cells), flush=True)
lib.me_debug_set_bf16_splitk(-1)
lib.me_debug_set_bf16_splitk_mode(0)
lib.me_debug_set_bf16_shape(0, 0)
