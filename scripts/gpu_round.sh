#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench, rocprofv3 kernel trace.  Everything is logged
# under gpurun_out/ (merged back by gpurun).  Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh [tag]'
set +e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== env" > $OUT/env.log
(rocm-smi --showproductname; nproc; lscpu | head -20; free -g) >> $OUT/env.log 2>&1
echo "== build+smoke"
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== pytest gpu"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu_x.log 2>&1; echo "pytest -x rc=$?" | tee -a $OUT/pytest_gpu_x.log
tail -5 $OUT/pytest_gpu_x.log
if ! grep -q " passed" $OUT/pytest_gpu_x.log || grep -q "failed" $OUT/pytest_gpu_x.log; then
  timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu_all.log 2>&1; echo "pytest all rc=$?" | tee -a $OUT/pytest_gpu_all.log
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_all.log | head -60
fi
echo "== bench"
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 300 python bench.py --steps 50 --warmup 10 --extent 215 --cpu-budget 5 > $OUT/bench_sparse.json 2> $OUT/bench_sparse.err
cut -c1-400 $OUT/bench_sparse.json
timeout 300 python bench.py --steps 50 --warmup 10 --dtype bf16 --cpu-budget 0 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
cut -c1-400 $OUT/bench_bf16.json
for dt in f32 bf16; do
  timeout 400 python bench.py --workload conv4d --steps 30 --warmup 5 --dtype $dt --cpu-budget 5 > $OUT/bench_conv4d_$dt.json 2> $OUT/bench_conv4d_$dt.err
  cut -c1-300 $OUT/bench_conv4d_$dt.json; echo
done
echo "== rocprofv3"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/prof.log 2>&1; echo "rocprof rc=$?"
cd $OLDPWD
find $OUT/prof -name "*stats*" | head; find $OUT/prof -name "*kernel_stats*" -exec head -25 {} \;
find $OUT/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep only the small summaries (the raw trace can be large)
find $OUT/prof -type f ! -name "*stats*" -size +2M -delete
echo "== done"
