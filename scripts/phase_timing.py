"""Per-phase cycle breakdown of k_conv_tile_f32<64,64> (instrumented build, variant 256; 272 = without gather)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
dev = torch.device("cuda:0")
lib = _lib.load()
NAMES = ["barrier A", "stage write+wait", "barrier B", "load issue", "multiply", "prologue", "epilogue", "batches"]
for extent in (70, 215):
    coords = make_scene(100000, extent, 0).to(dev)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    x = torch.rand(100000, 64, device=dev)
    w = torch.rand(27, 64, 128, device=dev) - 0.5
    for var in (256, 272):
        lib.me_debug_set_conv_variant(var)
        MEB._conv_forward(x, w, km, "mfma")
        torch.cuda.synchronize()
        lib.me_debug_conv_timing(None, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        MEB._conv_forward(x, w, km, "mfma")
        e.record()
        torch.cuda.synchronize()
        out = (ctypes.c_uint64 * 8)()
        lib.me_debug_conv_timing(out, 0)
        v = list(out)
        wgs = 2 * -(-100000 // MEB.plan_config(100000, 27, km.n_pairs, 64, 128)[0])
        tot = sum(v[:7])
        # s_memtime ticks at 100 MHz on gfx9 (constant clock): report shares and microseconds per workgroup
        print(f"extent {extent} var {var}: kernel {s.elapsed_time(e)*1e3:.0f} us, workgroups {wgs}, batches/wg {v[7]/wgs:.1f}")
        for n, c in zip(NAMES[:7], v[:7]):
            print(f"   {n:18s} {100.0*c/tot:5.1f} %   {c/wgs/100.0:8.2f} us per workgroup (100 MHz ticks)   {c/max(v[7],1)/100.0*1e3:7.1f} ns per batch")
    lib.me_debug_set_conv_variant(0)
