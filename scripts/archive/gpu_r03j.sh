#!/bin/bash
# Round-3 session J: two-deep row pipeline of k_wgrad_bf16 (tests + MinkUNet34C bf16 A/B), host time per layer on both
# hosts, the bf16 gradient error measurement, fused fp32 tests.
set +e
OUT=$PWD/gpurun_out/r03j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_conv.py tests/test_gpu_minkunet.py -m gpu -q --timeout 600 -x -k "two_steps or multi_offset or training_loss or minkunet14" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cat gpurun_out/config3_bf16_gradient_error.log 2>/dev/null | tail -2
for h in native python; do ME_AMD_HOST=$h timeout 200 python scripts/host_layer_time.py 2>/dev/null | grep host=; done
ME_AMD_HOST=python timeout 300 python scripts/unet_layers.py > $OUT/layers_deep.log 2>&1; grep "^step" $OUT/layers_deep.log
python - <<'PY' > gpurun_out/r03j/wgrad_ab.log 2>&1
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
for depth in (0, 1, 0, 1):
    lib.me_debug_set_wgrad_config(depth, 0)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"host {ME.get_host()} wgrad depth {'two steps in flight' if depth == 0 else 'one step'}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms / step")
PY
cat $OUT/wgrad_ab.log | grep -v amdgpu
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("$OUT/unet_bf16.json").read().strip().splitlines()[-1])
print("unet_bf16", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d.get("hip_graph"))
PY
