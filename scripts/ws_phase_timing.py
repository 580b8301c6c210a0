"""Where a batch's cycles go in k_conv_tile_bf16_ws: s_memtime phase counters of a -DME_WS_TIMING build
(ME_AMD_LIB_TAG=wst ME_AMD_EXTRA_HIPCC_FLAGS=-DME_WS_TIMING python -m minkowskiengine_amd.build), per layer shape:
cycles per batch of a multiplier wave (work between barriers / barrier wait) and of a producer wave, tile prologue and
epilogue.  s_memtime ticks at 100 MHz: x 24 = shader cycles at 2.4 GHz.
usage: ME_AMD_LIB_TAG=wst python scripts/ws_phase_timing.py  (GPU)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ["ME_AMD_HOST"] = "python"
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
import minkunet as MU
lib = _lib.load()
dev = torch.device("cuda:0")
cases = [("config 2", make_scene(100000, 70, 0).to(dev), 1, 64, 128)]
coords = MU.synthetic_scene(200000, seed=0).to(dev)
mgr0 = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr0.insert_and_map(coords, [1, 1, 1], "")
levels = {}
for ts in (2, 4, 8, 16):
    key = mgr0.stride(key, [2, 2, 2], "")
    levels[ts] = mgr0.get_coordinates(key).clone()
for ts, cin, cout in ((2, 96, 96), (4, 128, 128), (4, 64, 64), (4, 192, 128), (8, 128, 128), (8, 384, 256)):
    cases.append((f"unet ts{ts}", levels[ts], ts, cin, cout))
TICK = float(os.environ.get("TICK_CYCLES", "24"))   # shader cycles per s_memtime tick (100 MHz counter, 2.4 GHz clock)
print(f"{'case':>10s} {'layer':>10s} {'T':>4s} {'us':>7s} {'batches/tile':>12s} | multiplier: work  wait | producer: work  wait | prologue epilogue (cycles; per batch / per tile)")
for name, c, ts, cin, cout in cases:
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(c.shape[0], cin, generator=g) - 0.5).to(dev).bfloat16()
    w = (torch.rand(27, cin, cout, generator=g) - 0.5).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    k, _ = mgr.insert_and_map(c, [ts] * 3, "")
    km = mgr._kernel_map(k, k, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    for _ in range(3):
        MEB._conv_target(x, w, km, "out", km.n_out, name="conv_forward")
    torch.cuda.synchronize()
    lib.me_debug_ws_timing(None, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    s.record()
    for _ in range(reps):
        MEB._conv_target(x, w, km, "out", km.n_out, name="conv_forward")
    e.record()
    torch.cuda.synchronize()
    out = (ctypes.c_uint64 * 8)()
    lib.me_debug_ws_timing(out, 0)
    v = [int(t) for t in out]
    T = MEB._conv_launch_cfg(km, "out", km.n_out, cin, cout, True)[1][0]
    if v[4] == 0:
        print(f"{name:>10s} {str(cin) + '->' + str(cout):>10s} {T:4d}  (not on the wave-specialised kernel)")
        continue
    nb, nt = v[4], v[7]
    print(f"{name:>10s} {str(cin) + '->' + str(cout):>10s} {T:4d} {s.elapsed_time(e) / reps * 1e3:7.1f} {nb / nt:12.1f} | "
          f"{v[0] * TICK / nb:16.0f} {v[1] * TICK / nb:5.0f} | {v[2] * TICK / nb:14.0f} {v[3] * TICK / nb:5.0f} | "
          f"{v[5] * TICK / nt:8.0f} {v[6] * TICK / nt:8.0f}", flush=True)
