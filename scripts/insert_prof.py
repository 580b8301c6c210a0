"""Kernel-level profile of insert_and_map at N voxels (run under rocprofv3 --kernel-trace --stats): scripts/gpu_r06.sh insert_prof"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minkowskiengine_amd as ME   # noqa: E402
import bench                        # noqa: E402

n = int(os.environ.get("N", "100000"))
dev = torch.device("cuda:0")
coords = bench.make_scene(n, 70 if n <= 200000 else 200, 0).to(dev)
B = ME.host.backend()
for rep in range(int(os.environ.get("REPS", "20"))):
    mgr = B.CoordinateMapManagerGPU_c10()
    mgr.insert_and_map(coords, [1, 1, 1], "")
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ts = []
for rep in range(10):
    mgr = B.CoordinateMapManagerGPU_c10()
    torch.cuda.synchronize()
    ev[0].record()
    mgr.insert_and_map(coords, [1, 1, 1], "")
    ev[1].record()
    torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
print("insert_and_map us:", " ".join(f"{t:.1f}" for t in ts), "host", ME.get_host())
