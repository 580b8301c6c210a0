#!/bin/bash
# Round-4 session E: the driver's default bench run with in-run PMC traffic (wall time!), roctx marker trace of a
# fresh-scene MinkUNet34C step, the full GPU suite.
set +e
OUT=$PWD/gpurun_out/r04e
mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04e/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], "traffic", r["traffic"], r.get("traffic_over_compulsory"), r["traffic_note"][:80])
for k, v in d.get("workloads", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), "traffic", v.get("roofline", {}).get("traffic"), str(v.get("roofline", {}).get("traffic_note"))[:60], v.get("error"))
PY
cd /tmp
ME_AMD_ROCTX=1 timeout 600 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $OUT/roctx -o roctx -- python $GRAFT_REPO_ROOT/bench.py --workload minkunet --dtype bf16 --steps 3 --warmup 1 --cpu-budget 0 --extra-workloads off --pmc off --scenes fresh --min-time 0 --min-blocks 1 --max-blocks 1 --no-graph-probe > $OUT/roctx.log 2>&1; echo "roctx rc=$?"
cd $GRAFT_REPO_ROOT
find $OUT/roctx -name "*stats*.csv" | head; for f in $(find $OUT/roctx -name "*marker*stats*.csv" -o -name "*marker_api_stats.csv"); do echo "== $f"; head -12 $f; done
find $OUT/roctx -name "*.csv" -size +3M -delete
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
