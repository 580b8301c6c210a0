#!/bin/bash
set +e
TAG=${1:-r02v}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q --timeout 600 -s -k "vs_oracle or range_order or fixtures or full_size or reproducible" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|error relative" $OUT/pytest.log | head -20
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench.json 2>$OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("f32 cfg2", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["frac"])
PY
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --extent 215 > $OUT/bench_sparse.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench_sparse.json"))
print("f32 cfg2 sparse", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
timeout 600 python bench.py --workload minkunet --dtype f32 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_f32.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/unet_f32.json"))
print("f32 unet", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
