#!/bin/bash
# MinkUNet34C bench (f32 / bf16), torch-BN A/B, rocprofv3 kernel stats.  Usage: gpurun -- 'bash scripts/gpu_minkunet.sh [tag]'
set +e
TAG=${1:-r01_unet}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for dt in f32 bf16; do
  timeout 400 python bench.py --workload minkunet --steps 10 --warmup 3 --dtype $dt > $OUT/bench_minkunet_$dt.json 2> $OUT/bench_minkunet_$dt.err
  python -c "import json;d=json.load(open('$OUT/bench_minkunet_$dt.json'));print('$dt', d['ms_per_step'],'ms', d['value'],'Mpts/s', {k:(v['ms_per_step'],v['tflops']) for k,v in d['kernels'].items()})"
  ME_AMD_TORCH_BN=1 timeout 400 python bench.py --workload minkunet --steps 10 --warmup 3 --dtype $dt > $OUT/bench_minkunet_${dt}_torchbn.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/bench_minkunet_${dt}_torchbn.json'));print('$dt torch BN', d['ms_per_step'],'ms')"
done
cd /tmp
for dt in f32 bf16; do
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$dt -o trace -- python $OLDPWD/bench.py --workload minkunet --dtype $dt --steps 5 --warmup 2 > $OUT/prof_$dt.log 2>&1
find $OUT/prof_$dt -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_minkunet_$dt.csv \;
find $OUT/prof_$dt -type f ! -name "*stats*" -size +2M -delete
done
cd $OLDPWD
head -16 $OUT/kernel_stats_minkunet_bf16.csv | cut -c1-150
echo "== done"
