"""MinkUNet34C step time when the coordinate maps / kernel maps / plans are rebuilt every step (a new scene per
iteration, as in real training) vs with the maps cached."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
import minkunet as MU
dev = torch.device("cuda:0")
dt = torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32
coords = MU.synthetic_scene(200000, seed=0).to(dev)
feats = torch.rand(coords.shape[0], 3).to(dev).to(dt)
labels = torch.randint(0, 20, (coords.shape[0],)).to(dev)
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
crit = torch.nn.CrossEntropyLoss()
def step(x):
    opt.zero_grad(set_to_none=True)
    loss = crit(net(x).F.float(), labels)
    loss.backward()
    opt.step()
x = ME.SparseTensor(feats, coords)
for _ in range(3): step(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step(x)
torch.cuda.synchronize(); warm = (time.perf_counter() - t0) / 5
for _ in range(2): step(ME.SparseTensor(feats, coords))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step(ME.SparseTensor(feats, coords))     # new coordinate manager: every map rebuilt
torch.cuda.synchronize(); cold = (time.perf_counter() - t0) / 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    xs = ME.SparseTensor(feats, coords)
torch.cuda.synchronize(); ins = (time.perf_counter() - t0) / 5
print(f"maps cached: {warm*1e3:.1f} ms/step   maps rebuilt every step: {cold*1e3:.1f} ms/step   SparseTensor construction alone: {ins*1e3:.2f} ms")
