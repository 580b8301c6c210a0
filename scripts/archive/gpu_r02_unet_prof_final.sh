#!/bin/bash
# rocprofv3 kernel statistics of the MinkUNet34C step (bf16) on the final state
set +e
OUT=$PWD/gpurun_out/r02_unet_prof_final
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bf16 -o t -- python $REPO/bench.py --workload minkunet --dtype bf16 --steps 8 --warmup 2 --cpu-budget 0 --min-blocks 1 --min-time 0 > $OUT/bench_bf16.json 2> $OUT/prof_bf16.log
find $OUT/prof_bf16 -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_bf16.csv \;
find $OUT/prof_bf16 -type f ! -name "*stats*" -size +1M -delete
head -5 $OUT/kernel_stats_bf16.csv | cut -c1-120
