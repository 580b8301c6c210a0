#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r04g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_norm.py tests/test_gpu_bf16.py tests/test_gpu_native_host.py tests/test_gpu_minkunet.py -m gpu -q -x --timeout 900 > $OUT/pytest.log 2>&1
grep -v amdgpu.ids $OUT/pytest.log | tail -30
for s in 1 0; do
ME_AMD_CONV_BN_STATS=$s timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --no-graph-probe > $OUT/unet_bf16_$s.json 2> $OUT/unet_bf16_$s.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r04g/unet_bf16_$s.json').read().strip().split('\n')[-1]); print('unet bf16 stats=$s', d['ms_per_step'], d['config'].get('host_layer'))
except Exception as e: print('unreadable', e)
PY
done
