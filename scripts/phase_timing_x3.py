"""Per-phase cycle breakdown of k_conv_tile_f32x3 (instrumented build, debug variant 256), forward and dgrad of the
config-2 layer.  TILE env: tile rows (0 = plan config)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
dev = torch.device("cuda:0")
lib = _lib.load()
VAR = int(os.environ.get("VARIANT", "257"))   # 257: wave-specialised kernel, 256: ping-pong kernel
NAMES = (["multiplier barrier wait", "producer work", "producer barrier wait", "-", "multiplier work", "-", "-", "batches"]
         if VAR == 257 else ["barrier A", "split+stage write", "barrier B", "load issue", "multiply", "prologue", "epilogue", "batches"])
MEB._TILE_ROWS = int(os.environ.get("TILE", "0"))
coords = make_scene(100000, int(os.environ.get("EXTENT", "70")), 0).to(dev)
mgr = MEB.CoordinateMapManagerGPU_c10()
key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
x = torch.rand(100000, 64, device=dev)
gy = torch.rand(100000, 128, device=dev)
w = torch.rand(27, 64, 128, device=dev) - 0.5
for name, fn in (("forward 64->128", lambda: MEB._conv_forward(x, w, km, "mfma")),
                 ("dgrad 128->64", lambda: MEB._conv_target(gy, w, km, "in", km.n_in, name="conv_dgrad", transposed=True))):
    for var in (0 if VAR == 257 else 30, VAR):
        lib.me_debug_set_conv_variant(var)
        fn()
        torch.cuda.synchronize()
        lib.me_debug_conv_timing_f32x3(None, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if var in (0, 30):
            print(f"{name}: plain kernel {s.elapsed_time(e)*1e3:.0f} us")
            continue
        out = (ctypes.c_uint64 * 8)()
        lib.me_debug_conv_timing_f32x3(out, 0)
        v = list(out)
        tot = sum(v[:7])
        print(f"{name}: instrumented kernel {s.elapsed_time(e)*1e3:.0f} us, batches {v[7]}")
        for n, c in zip(NAMES[:7], v[:7]):
            print(f"   {n:18s} {100.0*c/max(tot,1):5.1f} %   {c/max(v[7],1)/100.0*1e3:7.1f} ns per batch (100 MHz ticks)")
lib.me_debug_set_conv_variant(0)
