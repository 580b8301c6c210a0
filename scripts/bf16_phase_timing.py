"""Phase counters of k_conv_tile_bf16 over one MinkUNet34C bf16 step (library built with -DME_BF16_TIMING:
ME_AMD_LIB_TAG=tim, scripts/ablate_bf16_tile.sh): where a workgroup's cycles go, deep pipeline and plain loop apart."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import minkowskiengine_amd as ME
from minkowskiengine_amd import _lib
import minkunet as MU
lib = _lib.load()
dev = torch.device("cuda:0")
coords = MU.synthetic_scene(200000, seed=0).to(dev)
x = ME.SparseTensor(torch.rand(coords.shape[0], 3).to(dev).bfloat16(), coords)
net = MU.MinkUNet34C(3, 20, D=3).to(dev)
labels = torch.randint(0, 20, (coords.shape[0],)).to(dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
def step():
    opt.zero_grad(set_to_none=True)
    MU.cross_entropy(net(x).F.float(), labels).backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 20)()
lib.me_debug_bf16_timing(None, 1)
step()
torch.cuda.synchronize()
lib.me_debug_bf16_timing(ctypes.cast(buf, ctypes.c_void_p), 0)
names = ["barrier A", "stage write + wait", "barrier B", "load issue", "multiply", "refill + descriptors", "prologue", "epilogue"]
for base, label in ((0, "deep pipeline (eight-wave workgroups)"), (10, "plain loop (four-wave workgroups)")):
    v = [int(buf[base + i]) for i in range(10)]
    tot = sum(v[:8])
    if tot == 0:
        print(label, ": no launches / library not built with -DME_BF16_TIMING"); continue
    print(f"{label}: {v[9]} workgroups, {v[8]} batches, {tot / 1e9:.2f} G cycles of wave-0 time; per batch {tot / max(v[8], 1):.0f} cycles")
    for n, c in zip(names, v[:8]):
        print(f"   {n:22s} {100.0 * c / tot:5.1f} %   {c / max(v[8], 1):7.1f} cycles per batch")
