#!/bin/bash
# GPU session for the bf16 path + config 5 + whole-network benches.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_bf16.sh [tag]'
set +e
TAG=${1:-r01_bf16}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest bf16 + config 5"
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_minkunet.py "tests/test_gpu_conv.py::test_config5_full_size" -m gpu -q --timeout 600 > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_new.log
grep -E "^(FAILED|ERROR)|passed|failed|Error|error" $OUT/pytest_new.log | head -40
echo "== bench conv3d f32 / bf16"
for dt in f32 bf16; do
  timeout 300 python bench.py --steps 50 --warmup 10 --dtype $dt --cpu-budget 3 > $OUT/bench_conv3d_$dt.json 2> $OUT/bench_conv3d_$dt.err; echo "rc=$?"
  cut -c1-1500 $OUT/bench_conv3d_$dt.json; tail -3 $OUT/bench_conv3d_$dt.err
done
echo "== bench conv4d"
for dt in f32 bf16; do
  timeout 400 python bench.py --workload conv4d --steps 30 --warmup 5 --dtype $dt --cpu-budget 5 > $OUT/bench_conv4d_$dt.json 2> $OUT/bench_conv4d_$dt.err; echo "rc=$?"
  cut -c1-1500 $OUT/bench_conv4d_$dt.json; tail -3 $OUT/bench_conv4d_$dt.err
done
echo "== bench minkunet"
for dt in f32 bf16; do
  timeout 400 python bench.py --workload minkunet --steps 10 --warmup 3 --dtype $dt > $OUT/bench_minkunet_$dt.json 2> $OUT/bench_minkunet_$dt.err; echo "rc=$?"
  cut -c1-2500 $OUT/bench_minkunet_$dt.json; tail -3 $OUT/bench_minkunet_$dt.err
done
echo "== rocprofv3 minkunet bf16"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --workload minkunet --dtype bf16 --steps 5 --warmup 2 > $OUT/prof.log 2>&1; echo "rocprof rc=$?"
cd $OLDPWD
find $OUT/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_minkunet_bf16.csv \;
head -22 $OUT/kernel_stats_minkunet_bf16.csv | cut -c1-160
find $OUT/prof -type f ! -name "*stats*" -size +2M -delete
echo "== done"
