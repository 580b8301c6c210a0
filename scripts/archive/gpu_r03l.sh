#!/bin/bash
# Round-3 session L: full GPU test suite after the reduce rewrite, then the step numbers.
set +e
OUT=$PWD/gpurun_out/r03l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16.json 2>/dev/null
timeout 600 python bench.py --workload minkunet --dtype f32 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_f32.json 2>/dev/null
timeout 300 python bench.py --cpu-budget 0 --extra-workloads off > $OUT/bench.json 2>/dev/null
timeout 300 python bench.py --workload conv4d --cpu-budget 0 > $OUT/conv4d.json 2>/dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"].get("frac"), (d.get("hip_graph") or {}).get("ms_per_step"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
