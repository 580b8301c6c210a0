#!/bin/bash
# last check of the round on the force-rebuilt libraries: full GPU suite, smoke, default line, MinkUNet34C bf16 line + stats
set +e
OUT=$PWD/gpurun_out/r03_final3
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 > $OUT/unet_bf16.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_unet -o trace -- python $REPO/bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --no-graph-probe > $OUT/prof_unet.json 2> $OUT/prof_unet.log
find $OUT/prof_unet -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_unet_bf16.csv \;
rm -rf $OUT/prof_unet
cd $REPO
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], "ms", d["config"].get("host_layer"), {k: v.get("ms_per_step") for k, v in d.get("workloads", {}).items()})
PY
