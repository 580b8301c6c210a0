"""cProfile of the host side of a MinkUNet34C bf16 training step (maps cached): where the Python time goes."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
import minkunet as MU
dev = torch.device("cuda:0")
dt = torch.bfloat16
coords = MU.synthetic_scene(200000, seed=0).to(dev)
x = ME.SparseTensor(torch.rand(coords.shape[0], 3).to(dev).to(dt), coords)
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
labels = torch.randint(0, 20, (coords.shape[0],)).to(dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
def fwd():
    return MU.cross_entropy(net(x).F.float(), labels)
def step():
    opt.zero_grad(set_to_none=True)
    fwd().backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
# enqueue time of the pieces (no sync inside; the queue is drained between measurements)
def T(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, r
with torch.no_grad():
    e, w, _ = T(fwd)
    print(f"forward, no_grad : enqueue {e:.2f} ms, wall {w:.2f} ms")
e, w, loss = T(fwd, 1)
print(f"forward with grad: enqueue {e:.2f} ms, wall {w:.2f} ms")
e, w, _ = T(lambda: loss.backward(), 1)
print(f"backward         : enqueue {e:.2f} ms, wall {w:.2f} ms")
e, w, _ = T(step, 10)
print(f"step             : enqueue {e:.2f} ms, wall {w:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(35)
