#!/bin/bash
# Round-3 session G: native-host test rerun, rocprofv3 kernel statistics of the MinkUNet34C bf16 step and of the default
# bench line (native host), default bench line with the workloads object.
set +e
OUT=$PWD/gpurun_out/r03g
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_native_host.py -m gpu -q --timeout 600 > $OUT/pytest_native.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_native.log
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; tail -4 $OUT/bench.err
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_unet -o trace -- python $REPO/bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/prof_unet.json 2> $OUT/prof_unet.log
find $OUT/prof_unet -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_unet_bf16.csv \;
find $OUT/prof_unet -type f ! -name "*stats*" -size +1M -delete
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o trace -- python $REPO/bench.py --cpu-budget 0 --extra-workloads off > $OUT/prof_bench.json 2> $OUT/prof_bench.log
find $OUT/prof_bench -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_bench.csv \;
find $OUT/prof_bench -type f ! -name "*stats*" -size +1M -delete
cd $REPO
python - <<PY
import json
for f in ("bench", "unet_bf16", "prof_unet", "prof_bench"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"].get("frac"), d["config"].get("host_layer"))
    for k, v in (d.get("workloads") or {}).items():
        print("  ", k, v.get("value"), v.get("ms_per_step"), v.get("wall_s"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
PY
head -30 $OUT/kernel_stats_unet_bf16.csv | cut -c1-150
