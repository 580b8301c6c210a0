"""Per-layer-shape convolution times inside a MinkUNet34C step (HIP-event timed): which shapes run far from the rest."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
from minkowskiengine_amd import backend as MEB
import minkunet as MU
dev = torch.device("cuda:0")
dt = torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32
coords = MU.synthetic_scene(200000, seed=0).to(dev)
x = ME.SparseTensor(torch.rand(coords.shape[0], 3).to(dev).to(dt), coords)
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
labels = torch.randint(0, 20, (coords.shape[0],)).to(dev)
crit = torch.nn.CrossEntropyLoss()
rec = collections.defaultdict(list)
orig_timed = MEB._timed
def timed(name, device, launch, flops=0.0):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); r = launch(); e.record()
    rec[(name, CUR[0])].append((s, e, flops))
    return r
CUR = [None]
orig_target, orig_bwd = MEB._conv_target, MEB._conv_backward
def target(src, kernel, km, tgt, n_tgt, name="conv_target", transposed=False):
    CUR[0] = (int(kernel.shape[0]), int(kernel.shape[1]), int(kernel.shape[2]), km.n_in, km.n_out, km.n_pairs)
    return orig_target(src, kernel, km, tgt, n_tgt, name=name, transposed=transposed)
def bwd(in_feat, grad_out, kernel, km, algo=None):
    CUR[0] = (int(kernel.shape[0]), int(kernel.shape[1]), int(kernel.shape[2]), km.n_in, km.n_out, km.n_pairs)
    return orig_bwd(in_feat, grad_out, kernel, km, algo)
def step():
    net.zero_grad(set_to_none=True)
    crit(net(x).F.float(), labels).backward()
for _ in range(3): step()
MEB._timed, MEB._conv_target, MEB._conv_backward = timed, target, bwd
for _ in range(3): step()
torch.cuda.synchronize()
rows = []
for (name, shp), evs in rec.items():
    t = sum(s.elapsed_time(e) for s, e, _ in evs) / 3
    fl = sum(f for _, _, f in evs) / 3
    rows.append((t, name, shp, len(evs) // 3, fl / (t * 1e-3) / 1e12 if t > 0 else 0))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"conv kernel time per step {tot:.2f} ms")
print(f"{'ms/step':>8s} {'calls':>5s} {'us/call':>8s} {'TF':>7s}  kernel        (K, Cin, Cout, n_in, n_out, pairs)")
for t, name, shp, calls, tf in rows[:30]:
    print(f"{t:8.3f} {calls:5d} {t/calls*1e3:8.1f} {tf:7.1f}  {name:12s} {shp}")
