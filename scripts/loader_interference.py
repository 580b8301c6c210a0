"""What a scene-building loader thread costs the training step (MinkUNet34C bf16, 200k voxels, cached maps):
ms per step of (a) eager launches and (b) a hipGraph replay of the step (no host work in the training thread), each
alone and beside a thread that keeps building new scenes (insert + recipe replay) on a high-priority stream.
(b) separates contention on the GPU from contention between the host threads (GIL, HIP runtime locks).
(A loader spinning in pure Python instead starves the training thread of the GIL outright — seconds per step.)
usage: python scripts/loader_interference.py  (GPU)"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
import minkunet as MU
dev = torch.device("cuda:0")
n = 200000
coords = MU.synthetic_scene(n, seed=0).to(dev)
g = torch.Generator().manual_seed(0)
feats = torch.rand(n, 3, generator=g).to(dev).bfloat16()
labels = torch.randint(0, 20, (n,), generator=g).to(dev)
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
x = ME.SparseTensor(feats, coords)

def step():
    opt.zero_grad(set_to_none=True)
    loss = MU.cross_entropy(net(x).F.float(), labels)
    loss.backward()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
ME.set_map_prefetch(True)
warm = ME.SparseTensor(feats, coords)       # (its manager replays x's recipe: complete)
torch.cuda.synchronize()

def timed(fn, steps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

stop = threading.Event()
built = [0]
def loader(mode):
    side = torch.cuda.Stream(priority=-1)
    torch.cuda.set_device(0)
    while not stop.is_set():
        with torch.cuda.stream(side):
            t = ME.SparseTensor(feats, coords)
        side.synchronize()
        built[0] += 1

def beside(fn, mode):
    stop.clear(); built[0] = 0
    th = threading.Thread(target=loader, args=(mode,), daemon=True)
    th.start()
    time.sleep(0.2)
    b0 = built[0]
    ms = timed(fn)
    nb = built[0] - b0
    stop.set(); th.join()
    return ms, nb

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
print(f"eager alone            {timed(step):7.2f} ms")
print(f"graph replay alone     {timed(graph.replay):7.2f} ms")
for mode in ("scenes",):
    ms, nb = beside(step, mode)
    print(f"eager  + loader({mode:6s}) {ms:7.2f} ms   ({nb} loader iterations in 20 steps)")
    ms, nb = beside(graph.replay, mode)
    print(f"replay + loader({mode:6s}) {ms:7.2f} ms   ({nb} loader iterations in 20 steps)")
