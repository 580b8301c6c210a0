#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r04n
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_norm.py tests/test_gpu_minkunet.py tests/test_gpu_native_host.py tests/test_gpu_distributed.py -m gpu -q -x --timeout 900 > $OUT/pytest.log 2>&1
grep -v amdgpu.ids $OUT/pytest.log | tail -3
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_unet -o trace -- python $REPO/bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --no-graph-probe > $OUT/prof_unet.json 2> $OUT/prof_unet.log
find $OUT/prof_unet -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_unet_bf16.csv \;
rm -rf $OUT/prof_unet
cd $REPO
grep "k_bn_final\|k_bn_bwd_final" $OUT/kernel_stats_unet_bf16.csv | cut -c1-120
timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --no-graph-probe > $OUT/unet_bf16.json 2> $OUT/unet_bf16.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04n/unet_bf16.json').read().strip().split('\n')[-1]); print('unet bf16', d['ms_per_step'])
PY
