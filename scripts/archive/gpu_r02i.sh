#!/bin/bash
set +e
TAG=${1:-r02i}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -30
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("conv3d", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["frac"], "cold", d["cold_ms"], d["cold"]["kernel_map_ms"], d["cold"]["plans_ms"])
PY
done
timeout 300 python bench.py --workload conv4d --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench4d.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench4d.json"))
print("conv4d", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["bound"], d["roofline"]["frac"], "cold", d["cold"])
PY
timeout 300 python bench.py --extent 215 --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_sparse.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench_sparse.json"))
print("sparse", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["bound"], d["roofline"]["frac"])
PY
