"""Wall time of CoordinateManager.prefetch on a new MinkUNet34C scene, by request kind (host-side, GPU idle before)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
from minkowskiengine_amd import backend as MEB
import minkunet as MU
dev = torch.device("cuda:0")
dt = torch.bfloat16
coords = MU.synthetic_scene(200000, seed=0).to(dev)
feats = torch.rand(coords.shape[0], 3).to(dev).to(dt)
labels = torch.randint(0, 20, (coords.shape[0],)).to(dev)
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
x = ME.SparseTensor(feats, coords)
MU.cross_entropy(net(x).F, labels).backward()
recipe = x.coordinate_manager.recipe()
print("requests:", collections.Counter(op[0] for op in recipe))
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = ME.SparseTensor(feats, coords)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    mgr = y.coordinate_manager._manager
    per = collections.defaultdict(float)
    for op in recipe:
        a = time.perf_counter()
        mgr.prefetch([op])
        per[op[0]] += time.perf_counter() - a
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"rep {rep}: insert {1e3*(t1-t0):.2f} ms, replay host {1e3*(t2-t1):.2f} ms (+ drain {1e3*(t3-t2):.2f}); "
          + ", ".join(f"{k} {1e3*v:.2f}" for k, v in per.items()))
if os.environ.get("PROFILE", "0") != "0":
    import cProfile, pstats
    y = ME.SparseTensor(feats, coords)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    y.coordinate_manager._manager.prefetch(recipe)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
