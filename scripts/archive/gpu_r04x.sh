#!/bin/bash
# Round-4 session X: the state after the wave-specialised kernel — full GPU suite, the driver's default bench run,
# MinkUNet34C bf16 (cached / fresh / loader thread), rocprofv3 kernel statistics of the cached step.
set +e
OUT=$PWD/gpurun_out/r04x
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04x/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], "traffic", r["traffic"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
for k, v in d.get("workloads", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), "frac", v.get("roofline", {}).get("frac"), "traffic", v.get("roofline", {}).get("traffic"), v.get("error"))
PY
run() {  # name, args...
  name=$1; shift
  timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 6 --cpu-budget 0 --pmc off "$@" > $OUT/unet_$name.json 2>$OUT/unet_$name.err
  python - <<PY
import json
d = json.loads(open("$OUT/unet_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"], d["config"].get("loader"))
PY
}
run cached
run fresh --scenes fresh
run fresh_replay --scenes fresh --replay-maps
run pipelined --scenes pipelined
run graph --graph
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o unet -- python $GRAFT_REPO_ROOT/bench.py --workload minkunet --dtype bf16 --steps 4 --warmup 3 --cpu-budget 0 --pmc off --min-time 0 --min-blocks 2 --max-blocks 2 > $OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_trace.csv" -delete
ls $OUT/prof
