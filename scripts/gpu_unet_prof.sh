#!/bin/bash
# rocprofv3 kernel statistics of the MinkUNet34C step (bench.py --workload minkunet), per dtype
set +e
TAG=${1:-unet_prof}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for dt in ${DTYPES:-bf16}; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$dt -o trace -- python $REPO/bench.py --workload minkunet --dtype $dt --steps 8 --warmup 2 --min-time 0 --max-blocks 1 --cpu-budget 0 > $OUT/bench_$dt.json 2> $OUT/prof_$dt.log
  find $OUT/prof_$dt -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_$dt.csv \;
  find $OUT/prof_$dt -type f ! -name "*stats*" -size +2M -delete
  python - <<P
import csv, json
rows = list(csv.DictReader(open("$OUT/kernel_stats_$dt.csv")))
steps = 11.0   # 1 cold + 2 warm-up + 8 timed
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("$dt: GPU kernel time per step %.2f ms" % (tot / steps / 1e6), "| bench:", json.load(open("$OUT/bench_$dt.json"))["ms_per_step"], "ms")
for r in rows[:32]:
    print(f"{float(r['TotalDurationNs'])/steps/1e3:8.1f} us/step {float(r['Percentage']):5.1f}% calls/step {int(r['Calls'])/steps:6.1f} avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:100]}")
P
done
