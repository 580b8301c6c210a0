#!/bin/bash
# Round-4 session O: what a loader thread costs the training thread — the loader builds scenes, the step trains on the
# cached scene (ME_BENCH_DISCARD_LOADED=1); kernel traces of cached and pipelined for the gap statistics
set +e
OUT=$PWD/gpurun_out/r04o
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, env..., -- args
  name=$1; shift
  timeout 300 env "$@" > $OUT/unet_$name.json 2>$OUT/unet_$name.err
  python - <<PY
import json
d = json.loads(open("$OUT/unet_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"], d["config"].get("loader"))
PY
}
B="python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 6 --cpu-budget 0 --pmc off"
run cached A=1 $B
run discard ME_BENCH_DISCARD_LOADED=1 $B --scenes pipelined
run pipelined A=1 $B --scenes pipelined
run discard_python_host ME_BENCH_DISCARD_LOADED=1 ME_AMD_HOST=python $B --scenes pipelined
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o cached -- python $GRAFT_REPO_ROOT/bench.py --workload minkunet --dtype bf16 --steps 4 --warmup 6 --cpu-budget 0 --pmc off --min-time 0 --min-blocks 2 --max-blocks 2 > $OUT/prof_cached.log 2>&1
