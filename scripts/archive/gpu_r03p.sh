#!/bin/bash
# deep (two batches ahead) pipeline of the eight-wave bf16 tile kernels: bit-identity tests, per-layer A/B inside a
# MinkUNet34C step (python host so the debug switch applies to the ctypes-loaded library), bench lines
set +e
OUT=$PWD/gpurun_out/r03p
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 > $OUT/pytest_bf16.log 2>&1
tail -3 $OUT/pytest_bf16.log
for deep in -1 1; do
  ME_AMD_HOST=python BF16_DEEP=$deep timeout 300 python scripts/unet_layers.py > $OUT/layers_deep_$deep.log 2>&1
  grep "^step" $OUT/layers_deep_$deep.log
done
timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --no-graph-probe > $OUT/unet_bf16.json 2> $OUT/unet_bf16.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03p/unet_bf16.json').read().strip().split('\n')[-1]); print('unet bf16', d['ms_per_step'], d['config'].get('host_layer'))
except Exception as e: print('unreadable', e)
PY
