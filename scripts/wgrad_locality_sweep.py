"""Weight gradient of the headline layer (100k voxels in 70^3, 64 -> 128): us per launch (kernel + reduce) for row-ordered
(flat-table map) and position-ordered (spatial map) pair lists, by the number of pair ranges (workgroups per CU).
usage: python scripts/wgrad_locality_sweep.py  (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ME_AMD_HOST", "python")
import torch
from minkowskiengine_amd import backend as MEB, _lib
from bench import make_scene
lib = _lib.load()
dev = torch.device("cuda:0")
DT = torch.bfloat16 if os.environ.get("DTYPE", "f32") == "bf16" else torch.float32
coords = make_scene(100000, 70, 0).to(dev)
x = torch.rand(100000, 64, device=dev).to(DT)
gy = torch.rand(100000, 128, device=dev).to(DT)
w = torch.rand(27, 64, 128, device=dev) - 0.5
print(f"{'pair lists':>16s} " + " ".join(f"{'wpc ' + str(p):>10s}" for p in (0, 2, 3, 4)))
for spatial in (False, True):
    MEB._SPATIAL_MAPS = spatial
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    cells = []
    for wpc in (0, 2, 3, 4):
        lib.me_debug_set_wgrad_config(0, wpc)
        km._launch_cache.clear()
        MEB._WGRAD_TUNING = True
        for _ in range(3):
            MEB._conv_backward(x, gy, w, km, "mfma", need_grad_in=False)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            MEB._conv_backward(x, gy, w, km, "mfma", need_grad_in=False)
        e.record()
        torch.cuda.synchronize()
        cells.append(f"{s.elapsed_time(e) / 20 * 1e3:8.1f}")
    print(f"{'position order' if spatial else 'row order':>16s} " + " ".join(f"{c:>10s}" for c in cells), flush=True)
lib.me_debug_set_wgrad_config(0, 0)
