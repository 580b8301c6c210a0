#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03r
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_norm.py tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
ME_AMD_HOST=python timeout 300 python scripts/bn_bandwidth.py > $OUT/bn_bandwidth.log 2>&1
grep -v amdgpu.ids $OUT/bn_bandwidth.log
timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --no-graph-probe > $OUT/unet_bf16.json 2> $OUT/unet_bf16.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03r/unet_bf16.json').read().strip().split('\n')[-1]); print('unet bf16', d['ms_per_step'], d['config'].get('host_layer'))
except Exception as e: print('unreadable', e)
PY
