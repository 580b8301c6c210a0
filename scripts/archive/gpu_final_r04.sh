#!/bin/bash
# Round-4 closing session: the driver's sequence on one box — build check (prebuilt libraries load), GPU suite, smoke,
# default bench (with in-run PMC), kernel statistics of the default command.
set +e
OUT=$PWD/gpurun_out/r04_final
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_final/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
for k, v in d.get("workloads", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), "frac", v.get("roofline", {}).get("frac"), "traffic", v.get("roofline", {}).get("traffic"), v.get("error"))
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o default -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-budget 0 --extra-workloads off --pmc off > $OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_trace.csv" -delete
ls $OUT/prof
