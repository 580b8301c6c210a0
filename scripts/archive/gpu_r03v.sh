#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03v
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_norm.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1
tail -2 $OUT/pytest.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_unet -o trace -- python $REPO/bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --no-graph-probe > $OUT/prof_unet.json 2> $OUT/prof_unet.log
find $OUT/prof_unet -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_unet_bf16.csv \;
rm -rf $OUT/prof_unet
cd $REPO
timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --no-graph-probe > $OUT/unet_bf16.json 2> $OUT/unet_bf16.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03v/unet_bf16.json').read().strip().split('\n')[-1]); print('unet bf16', d['ms_per_step'], d['config'].get('host_layer'))
except Exception as e: print('unreadable', e)
PY
