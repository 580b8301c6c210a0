#!/bin/bash
# Where does the step time outside the three convolution kernels go with position-space maps?  (bf16, config 2)
set +e
OUT=$PWD/gpurun_out/r02_exp10
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for mode in rows spatial; do
  if [ $mode = spatial ]; then export ME_AMD_SPATIAL_MAPS=1 ME_AMD_TILE_ORDER=spatial; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$mode -o t -- python $REPO/bench.py --cpu-budget 0 --dtype bf16 --steps 50 --warmup 10 --min-blocks 1 --min-time 0 > $OUT/bench_$mode.json 2> $OUT/prof_$mode.log
  find $OUT/prof_$mode -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_$mode.csv \;
  find $OUT/prof_$mode -type f ! -name "*stats*" -size +1M -delete
done
cd $REPO
for mode in rows spatial; do echo "== $mode"; head -14 $OUT/kernel_stats_$mode.csv | cut -c1-150; done
