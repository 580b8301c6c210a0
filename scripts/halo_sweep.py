"""The halo kernel (csrc/conv_halo.hip) against the tile-plan kernels on every level of the MinkUNet34C scene: us per forward
launch per channel shape, for the configurations in CONFIGS ("off" = the shipped tile-plan policy; "T,kc,skip" = forced
halo kernel with that tile height / channel chunk (0 = its default) / empty-group skip), and the largest element-wise
difference of each halo result from the tile-plan result in units of the bf16 tolerance of tests/test_gpu_bf16.py.
usage: python scripts/halo_sweep.py   (GPU; LEVELS=1,2,4,8,16  CONFIGS="off;128,0,1;64,0,1;128,0,0"  REPS=20)"""
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
levels = {1: coords}
key = k1
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
ALL = {1: [(96, 96), (128, 96)], 2: [(96, 96), (32, 32), (128, 96)], 4: [(128, 128), (64, 64), (192, 128), (32, 64)],
       8: [(128, 128), (256, 256), (384, 256), (64, 128)], 16: [(256, 256), (128, 256)]}
if os.environ.get("DGRAD_SHAPES", "0") == "1":     # the input-gradient launches of the same layers: (Cout, Cin) as (src, dst)
    ALL = {ts: [(b, a) for a, b in v if a != b] for ts, v in ALL.items()}
want = os.environ.get("LEVELS", "1,2,4,8,16")
CONFIGS = os.environ.get("CONFIGS", "off;128,0,1;64,0,1;128,0,0").split(";")
REPS = int(os.environ.get("REPS", "20"))
TIMING = os.environ.get("TIMING", "0") == "1"    # -DME_HALO_TIMING build: phase counters per configuration
TARGET = os.environ.get("TARGET", "fwd")      # fwd | bwd (input gradient + weight gradient through _conv_backward)
print(f"{'level':>6s} {'rows':>7s} {'cin->cout':>10s} " + " ".join(f"{c:>16s}" for c in CONFIGS))
for ts in [int(l) for l in want.split(",")]:
    c = levels[ts]
    for cin, cout in ALL[ts]:
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
        w = (torch.rand(27, cin, cout, generator=torch.Generator().manual_seed(2)) - 0.5).to(dev)
        gy = (torch.rand(c.shape[0], cout, generator=g) - 0.5).to(dev).bfloat16()
        cells, ref = [], None
        seen_plans = set()
        for cfg in CONFIGS:
            if cfg == "off":
                lib.me_debug_set_halo(0, 0, 0, 1)
            else:
                t, kc, skip = (int(v) for v in cfg.split(","))
                lib.me_debug_set_halo(1, t, kc, skip)
            mgr = MEB.CoordinateMapManagerGPU_c10()
            key, _ = mgr.insert_and_map(c, [ts] * 3, "")
            km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
            run = (lambda: MEB._conv_forward(x, w, km, "mfma")) if TARGET == "fwd" else \
                (lambda: MEB._conv_backward(x, gy, w, km, "mfma")[0])
            try:
                for _ in range(3):
                    y = run()
                torch.cuda.synchronize()
            except Exception as e:       # a shape without an instantiation
                cells.append("n/a")
                continue
            if TIMING and cfg != "off":
                lib.me_debug_halo_timing(None, 1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(REPS):
                run()
            e.record()
            torch.cuda.synchronize()
            if cfg != "off" and os.environ.get("HALO_STATS", "1") == "1":
                for nm, v in km._store.items():
                    if isinstance(nm, str) and "halo" in nm and v is not None and nm not in seen_plans:
                        seen_plans.add(nm)
                        cnt = v[2].float()
                        print(f"        {nm}: tiles {cnt.numel()} halo mean {cnt.mean().item():.0f} max {cnt.max().item():.0f} "
                              f"over cap {int((cnt > v[1]).sum())} native_order {km.table_pos('out')[1] is not None}", flush=True)
            if TIMING and cfg != "off":
                import ctypes
                buf = (ctypes.c_uint64 * 8)()
                lib.me_debug_halo_timing(buf, 0)
                tiles = max(1, buf[4])
                print(f"        [{cfg}] ticks/tile: prologue {buf[0] / tiles:.0f} stage {buf[1] / tiles:.0f} walk {buf[2] / tiles:.0f} "
                      f"epilogue {buf[3] / tiles:.0f}; walk ticks/offset {buf[2] / max(1, buf[5]):.0f} (offsets/tile {buf[5] / tiles:.1f})", flush=True)
            yf = y.float()
            if ref is None:
                ref = yf
                tag = ""
            else:
                tol = 2.0 ** -8 * ref.abs() + 1e-3 * max(1.0, float(ref.abs().max()))
                tag = f" d{float(((yf - ref).abs() / tol).max()):.2f}"
            cells.append(f"{s.elapsed_time(e) / REPS * 1e3:7.1f}{tag}")
        print(f"{ts:6d} {c.shape[0]:7d} {str(cin) + '->' + str(cout):>10s} " + " ".join(f"{v:>16s}" for v in cells), flush=True)
lib.me_debug_set_halo(-1, 0, 0, 1)
