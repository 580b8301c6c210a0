"""MinkUNet34C steps with every coordinate map / kernel map / tile plan rebuilt (MODE=cold, a new scene per
iteration as in training) or cached (MODE=warm); run under rocprofv3 --kernel-trace --stats to split the cold
overhead into map-kernel GPU time and host-synchronisation bubbles.  Prints wall ms/step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
import minkunet as MU
dev = torch.device("cuda:0")
mode = os.environ.get("MODE", "cold")
steps = int(os.environ.get("STEPS", "10"))
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
for _ in range(3):
    step(x if mode == "warm" else ME.SparseTensor(feats, coords))
if os.environ.get("PROFILE", "0") != "0":      # host-side picture of the cold path (cProfile inflates everything ~2x)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step(x if mode == "warm" else ME.SparseTensor(feats, coords))
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(30)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    step(x if mode == "warm" else ME.SparseTensor(feats, coords))
torch.cuda.synchronize()
print(f"MODE={mode} dtype={dt} wall {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step over {steps} steps")
