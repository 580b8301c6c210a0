#!/bin/bash
set +e
TAG=${1:-r02d}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_coords.py -m gpu -q --timeout 600 -x > $OUT/pytest_coords.log 2>&1; echo "coords rc=$?"
tail -25 $OUT/pytest_coords.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -30
echo "== bench cold numbers"
for w in conv3d conv4d; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?"
  python - <<PY
import json
d = json.load(open("$OUT/bench_$w.json"))
print("$w", d["value"], d["ms_per_step"], d["cold"], d["kernels"])
PY
done
ME_AMD_SPATIAL_MAPS=0 timeout 300 python bench.py --workload conv3d --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_conv3d_flat.json 2> /dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench_conv3d_flat.json"))
print("conv3d flat", d["value"], d["ms_per_step"], d["cold"], d["kernels"])
PY
echo "== done"
