"""What the gradient exchange machinery costs a MinkUNet34C bf16 step on ONE rank (a one-rank RCCL group: the collectives
run, nobody answers): plain step | torch DDP (25 MB buckets) | DDP (one bucket) | flat all-reduce after backward
(distributed.allreduce_gradients) | gradient arena (distributed.GradientArena).  usage: python scripts/ddp_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import torch
import minkowskiengine_amd as ME
from minkowskiengine_amd import distributed as D
import minkunet as MU
D.init_from_env(backend="nccl")
dev = torch.device("cuda:0")
coords = MU.synthetic_scene(200000, seed=0)
g = torch.Generator().manual_seed(0)
feats = torch.rand(coords.shape[0], 3, generator=g)
labels = torch.randint(0, 20, (coords.shape[0],), generator=g).to(dev)
x = ME.SparseTensor(feats.to(dev).bfloat16(), coords.to(dev))
MODES = os.environ.get("MODES", "plain,ddp25,ddp1000,flat,arena,arena4").split(",")
STEPS = int(os.environ.get("STEPS", "20"))
for mode in MODES:
    torch.manual_seed(0)
    model = MU.MinkUNet34C(3, 20, D=3).to(dev).train()
    net, arena = model, None
    if mode.startswith("ddp"):
        net = D.data_parallel(model, dev, bucket_cap_mb=int(mode[3:]))
    if mode.startswith("arena"):
        arena = D.GradientArena(model, chunks=int(mode[5:] or 1))
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)

    def step():
        if arena is not None:
            arena.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        loss = MU.cross_entropy(net(x).F, labels)
        loss.backward()
        if mode == "flat":
            D.allreduce_gradients(model)
        if arena is not None:
            arena.all_reduce()
        opt.step()
    for _ in range(5):
        step()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / STEPS * 1e3)
    print(f"{mode:8s} ms/step median {sorted(ts)[2]:.3f}  blocks {' '.join(f'{t:.3f}' for t in ts)}" +
          (f"  arena: {arena.describe()}" if arena is not None else ""), flush=True)
    del net, model, opt, arena
D.shutdown()
