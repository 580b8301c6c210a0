#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r04s
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1
grep -v amdgpu.ids $OUT/pytest.log | grep -v "^  File" | tail -3
ME_AMD_HOST=python timeout 300 python scripts/unet_layers.py > $OUT/layers.log 2>&1
grep "^step" $OUT/layers.log
timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --no-graph-probe > $OUT/unet_bf16.json 2> $OUT/unet_bf16.err
timeout 300 python bench.py --cpu-budget 0 --extra-workloads off > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
for n in ('unet_bf16','bench'):
    try:
        d=json.loads(open(f'gpurun_out/r04s/{n}.json').read().strip().split('\n')[-1]); print(n, d['value'], d['ms_per_step'], {k: round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(n,'unreadable', e)
PY
