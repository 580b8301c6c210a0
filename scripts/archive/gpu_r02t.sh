#!/bin/bash
set +e
TAG=${1:-r02t}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_norm.py tests/test_gpu_conv.py tests/test_gpu_bf16.py tests/test_gpu_coords.py tests/test_gpu_minkunet.py -m gpu -q --timeout 600 -k "not lds_dma" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -20
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench.json 2>$OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("f32 cfg2", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["frac"])
PY
for dt in bf16 f32; do
timeout 600 python bench.py --workload minkunet --dtype $dt --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_$dt.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/unet_$dt.json"))
print("$dt unet", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --graph > $OUT/unet_bf16_graph.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/unet_bf16_graph.json"))
print("bf16 unet graph", d["value"], d["ms_per_step"])
PY
for m in warm cold; do MODE=$m timeout 300 python scripts/cold_profile.py 2>&1 | tail -1; done
