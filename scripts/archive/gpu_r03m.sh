#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03m
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_unet -o trace -- python $REPO/bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --no-graph-probe > $OUT/prof_unet.json 2> $OUT/prof_unet.log
find $OUT/prof_unet -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_unet_bf16.csv \;
find $OUT/prof_unet -type f ! -name "*stats*" -size +1M -delete
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o trace -- python $REPO/bench.py --cpu-budget 0 --extra-workloads off > $OUT/prof_bench.json 2> $OUT/prof_bench.log
find $OUT/prof_bench -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_bench.csv \;
find $OUT/prof_bench -type f ! -name "*stats*" -size +1M -delete
cd $REPO
head -42 $OUT/kernel_stats_unet_bf16.csv | cut -c1-150
