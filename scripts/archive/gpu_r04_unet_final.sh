#!/bin/bash
# Round-4: MinkUNet34C bf16 on the last build — kernel statistics of the cached step, per-layer table, the step with
# cached maps / a new scene every step / loader thread / hipGraph
set +e
OUT=$PWD/gpurun_out/r04_unet_final
mkdir -p $OUT
export TMPDIR=/tmp
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
ME_AMD_HOST=python python scripts/unet_layers.py > $OUT/layers.log 2>&1; head -3 $OUT/layers.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o unet -- python $GRAFT_REPO_ROOT/bench.py --workload minkunet --dtype bf16 --steps 4 --warmup 3 --cpu-budget 0 --pmc off --min-time 0 --min-blocks 2 --max-blocks 2 > $OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_trace.csv" -delete
