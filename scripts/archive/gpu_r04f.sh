#!/bin/bash
# Round-4 session F: TWOBUF (two stage buffers, one barrier per batch) of the deep-pipeline bf16 kernels: bit-identity + sweep + step.
set +e
OUT=$PWD/gpurun_out/r04f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -x -k "deep_pipeline or split_k" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
KNOB=twobuf MODES=0,1 LEVELS=4,8,16 timeout 600 python scripts/offsync_sweep.py > $OUT/twobuf_sweep.log 2>&1; grep -v amdgpu $OUT/twobuf_sweep.log
python - <<'PY' > gpurun_out/r04f/unet_twobuf.log 2>&1
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "examples"))
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
for two in (0, 1, 0, 1):
    lib.me_debug_set_bf16_twobuf(two)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"host {ME.get_host()} twobuf {two}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms / step")
PY
grep -v amdgpu gpurun_out/r04f/unet_twobuf.log
