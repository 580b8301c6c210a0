#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03u
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_norm.py tests/test_gpu_minkunet.py tests/test_gpu_native_host.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
cd /tmp
ME_AMD_HOST=python timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o bn -- python $REPO/scripts/bn_bandwidth.py > $OUT/bn_bandwidth.log 2>&1
cd $REPO
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python scripts/trace_by_grid.py $f k_bn > $OUT/bn_by_grid.log 2>&1
cat $OUT/bn_by_grid.log
rm -rf $OUT/prof
timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --no-graph-probe > $OUT/unet_bf16.json 2> $OUT/unet_bf16.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03u/unet_bf16.json').read().strip().split('\n')[-1]); print('unet bf16', d['ms_per_step'], d['config'].get('host_layer'))
except Exception as e: print('unreadable', e)
PY
