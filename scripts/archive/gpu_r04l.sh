#!/bin/bash
# Round-4 session L: me_plan_build_multi — tests, then MinkUNet34C bf16 with a new scene every step: lazy builds, the
# recipe replay (all plans in four launches), pipelined; cached for reference.
set +e
OUT=$PWD/gpurun_out/r04l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_coords.py tests/test_gpu_prefetch.py tests/test_gpu_native_host.py tests/test_gpu_bf16.py -x -q -m gpu -k "plan or prefetch or recipe or tile_order" 2>&1 | tail -5 | tee $OUT/pytest.log
run() {  # name, args...
  name=$1; shift
  python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --pmc off "$@" > $OUT/unet_$name.json 2>$OUT/unet_$name.err
  python - <<PY
import json
d = json.loads(open("$OUT/unet_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"], d["config"].get("map_prefetch"))
PY
}
run cached
run fresh_lazy --scenes fresh --lazy-maps
run fresh --scenes fresh
run pipelined --scenes pipelined
