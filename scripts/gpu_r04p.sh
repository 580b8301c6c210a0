#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r04p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_scene_prefetch.py tests/test_gpu_prefetch.py -q -m gpu 2>&1 | tail -3
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
run graph A=1 $B --graph
