#!/bin/bash
# Round-4 session N: the loader thread — high-priority stream, depth 1 / 2; what the loader and the consumer wait for
set +e
OUT=$PWD/gpurun_out/r04n
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
run pipelined_d1 --scenes pipelined
run pipelined_d2 --scenes pipelined --loader-depth 2
run cached
cd /tmp
ME_AMD_ROCTX=1 timeout 300 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $OUT/prof -o pipe -- python $GRAFT_REPO_ROOT/bench.py --workload minkunet --dtype bf16 --steps 4 --warmup 6 --cpu-budget 0 --pmc off --scenes pipelined --min-time 0 --min-blocks 2 --max-blocks 2 > $OUT/prof.log 2>&1
ls $OUT/prof | head
