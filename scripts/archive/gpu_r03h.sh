#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "
import torch, sys
sys.path.insert(0, '.')
import minkowskiengine_amd as ME
print('host', ME.get_host())
from tests.helpers import make_cloud
c = make_cloud(3000, 14, 3, seed=1).cuda()
x = ME.SparseTensor(torch.rand(c.shape[0], 16).cuda(), c)
y = ME.MinkowskiConvolution(16, 32, kernel_size=3, dimension=3).cuda()(x)
torch.cuda.synchronize(); print('ok', y.F.shape)
" > $OUT/quick.log 2>&1; tail -3 $OUT/quick.log
timeout 300 python -m pytest tests/test_gpu_native_host.py tests/test_gpu_bf16.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 200 python bench.py --cpu-budget 0 --extra-workloads off > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16.json 2>$OUT/unet.err; echo "unet rc=$?"
ME_AMD_HOST=python timeout 300 python scripts/unet_layers.py > $OUT/layers_new.log 2>&1; grep "^step" $OUT/layers_new.log
python - <<PY
import json
for f in ("bench", "unet_bf16"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d.get("hip_graph"), d.get("cold"))
    except Exception as e:
        print(f, "unreadable", e)
PY
