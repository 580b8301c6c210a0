"""The two bf16 forward schedules on every level of the MinkUNet34C scene: column-split k_conv_tile_bf16 (mode 0) against
the offset-synchronous k_conv_off_bf16 in its two wave shapes (modes 1 / 2): us per forward launch, per channel shape.
usage: python scripts/offsync_sweep.py  (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ.setdefault("ME_AMD_HOST", "python")
import torch
from minkowskiengine_amd import backend as MEB, _lib
import minkunet as MU
lib = _lib.load()
lib.me_debug_set_bf16_splitk(0)
dev = torch.device("cuda:0")
coords = MU.synthetic_scene(200000, seed=0).to(dev)
mgr0 = MEB.CoordinateMapManagerGPU_c10()
k1, _ = mgr0.insert_and_map(coords, [1, 1, 1], "")
levels = {1: coords}
key = k1
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
ALL = {1: [(96, 96), (128, 96)], 2: [(96, 96), (32, 32), (128, 96)], 4: [(128, 128), (64, 64), (192, 128), (32, 64)],
       8: [(128, 128), (256, 256), (384, 256), (64, 128)], 16: [(256, 256), (128, 256)]}
want = os.environ.get("LEVELS", "1,2,4,8,16")
MODES = [int(m) for m in os.environ.get("MODES", "0,1,2").split(",")]
KNOB = os.environ.get("KNOB", "offsync")      # which switch the modes set: offsync | twobuf | deep
SETTER = {"offsync": lib.me_debug_set_bf16_offsync, "twobuf": lib.me_debug_set_bf16_twobuf,
          "deep": lib.me_debug_set_bf16_deep}[KNOB]
REPS = int(os.environ.get("REPS", "20"))
print(f"{'level':>6s} {'rows':>7s} {'cin->cout':>10s} " + " ".join(f"{KNOB + ' ' + str(m):>14s}" for m in MODES))
for ts in [int(l) for l in want.split(",")]:
    c = levels[ts]
    for cin, cout in ALL[ts]:
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
        cells, outs = [], []
        for mode in MODES:
            SETTER(mode)
            w = (torch.rand(27, cin, cout, generator=torch.Generator().manual_seed(2)) - 0.5).to(dev)
            mgr = MEB.CoordinateMapManagerGPU_c10()
            key, _ = mgr.insert_and_map(c, [ts] * 3, "")
            km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
            T, _, sk = MEB.plan_config(km.n_out, km.volume, km.n_pairs, cin, cout, True, False, with_split_k=True)
            for _ in range(3):
                y = MEB._conv_forward(x, w, km, "mfma")
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(REPS):
                MEB._conv_forward(x, w, km, "mfma")
            e.record()
            torch.cuda.synchronize()
            outs.append(y.float().abs().sum().item())
            cells.append(f"{s.elapsed_time(e) / REPS * 1e3:7.1f} T{T:<3d}")
        same = all(abs(o - outs[0]) <= 1e-3 * abs(outs[0]) for o in outs)
        print(f"{ts:6d} {c.shape[0]:7d} {str(cin) + '->' + str(cout):>10s} " + " ".join(f"{v:>14s}" for v in cells) +
              ("" if same else "   CHECKSUM MISMATCH"), flush=True)
SETTER(0 if KNOB == "offsync" else -1)
lib.me_debug_set_bf16_splitk(-1)
