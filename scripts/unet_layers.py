"""Per-layer-shape convolution times inside a MinkUNet34C step (HIP-event timed): which shapes run far from the rest.
Prints every (kernel, layer shape) row with calls, us per call, TFLOP/s, tile geometry, and the step time."""
import os, sys, collections, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ.setdefault("ME_AMD_HOST", "python")     # (the table is recorded by patching the ctypes host's launch functions)
import torch
import minkowskiengine_amd as ME
from minkowskiengine_amd import backend as MEB
import minkunet as MU
if os.environ.get("BF16_SHAPE"):     # "nc,kc": tuning override of the bf16 tile kernel's slab width / chunk depth
    from minkowskiengine_amd import _lib
    _lib.load().me_debug_set_bf16_shape(*[int(v) for v in os.environ["BF16_SHAPE"].split(",")])
if os.environ.get("BF16_WS_DEPTH"):  # 2 / 4: register sets of gathered rows in flight in the wave-specialised bf16 kernel's producers
    from minkowskiengine_amd import _lib
    _lib.load().me_debug_set_bf16_ws_depth(int(os.environ["BF16_WS_DEPTH"]))
if os.environ.get("WGRAD_WPC"):      # workgroups per CU the weight-gradient ranges are sized for (0 = policy)
    from minkowskiengine_amd import _lib
    _lib.load().me_debug_set_wgrad_config(0, int(os.environ["WGRAD_WPC"]))
if os.environ.get("WGRAD_MB"):       # 4 / 8: input-channel blocks per workgroup of k_wgrad_bf16 (0 = policy)
    from minkowskiengine_amd import _lib
    _lib.load().me_debug_set_wgrad_mb(int(os.environ["WGRAD_MB"]))
if os.environ.get("BF16_DEEP"):      # -1 policy / 0 never / 1 wherever instantiated: deep pipeline of the eight-wave kernels
    from minkowskiengine_amd import _lib
    _lib.load().me_debug_set_bf16_deep(int(os.environ["BF16_DEEP"]))
if os.environ.get("BF16_SPLITK"):    # -1 policy / 0 never / G: offset groups of the split-K launches (small maps)
    from minkowskiengine_amd import _lib
    _lib.load().me_debug_set_bf16_splitk(int(os.environ["BF16_SPLITK"]))
dev = torch.device("cuda:0")
dt = torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32
coords = MU.synthetic_scene(200000, seed=0).to(dev)
x = ME.SparseTensor(torch.rand(coords.shape[0], 3).to(dev).to(dt), coords)
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
labels = torch.randint(0, 20, (coords.shape[0],)).to(dev)
crit = MU.cross_entropy
opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
rec = collections.defaultdict(list)
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
def bwd(in_feat, grad_out, kernel, km, algo=None, **kw):
    CUR[0] = (int(kernel.shape[0]), int(kernel.shape[1]), int(kernel.shape[2]), km.n_in, km.n_out, km.n_pairs)
    return orig_bwd(in_feat, grad_out, kernel, km, algo, **kw)
def step():
    opt.zero_grad(set_to_none=True)
    crit(net(x).F.float(), labels).backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
step_ms = (time.perf_counter() - t0) / 5 * 1e3
MEB._timed, MEB._conv_target, MEB._conv_backward = timed, target, bwd
MEB.KERNEL_TIMER = True      # (the weight-gradient launch is timed only with a timer installed)
for _ in range(3): step()
torch.cuda.synchronize()
rows = []
for (name, shp), evs in rec.items():
    t = sum(s.elapsed_time(e) for s, e, _ in evs) / 3
    fl = sum(f for _, _, f in evs) / 3
    K, ci, co, n_in, n_out, P = shp
    if name == "conv_wgrad":
        geo = ""
    else:
        cs, cd, nt = (co, ci, n_in) if name == "conv_dgrad" else (ci, co, n_out)
        T, G = MEB.plan_config(nt, K, P, cs, cd, dt == torch.bfloat16, False)
        geo = f"T={T} tiles={-(-nt // T)}"
    rows.append((t, name, shp, len(evs) // 3, fl / (t * 1e-3) / 1e12 if t > 0 else 0, geo))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"step {step_ms:.2f} ms; conv kernel time per step {tot:.2f} ms "
      f"(fwd {sum(r[0] for r in rows if r[1]=='conv_forward'):.2f} dgrad {sum(r[0] for r in rows if r[1]=='conv_dgrad'):.2f} "
      f"wgrad {sum(r[0] for r in rows if r[1]=='conv_wgrad'):.2f})")
print(f"{'ms/step':>8s} {'calls':>5s} {'us/call':>8s} {'TF':>7s}  kernel        (K, Cin, Cout, n_in, n_out, pairs)")
for t, name, shp, calls, tf, geo in rows:
    print(f"{t:8.3f} {calls:5d} {t/calls*1e3:8.1f} {tf:7.1f}  {name:12s} {shp} {geo}")
