#!/bin/bash
set +e
TAG=${1:-r02j}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_minkunet.py tests/test_gpu_conv.py -m gpu -q --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -30
for g in 1 0; do
  ME_AMD_BF16_GATHER=$g timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --dtype bf16 > $OUT/bench_bf16_$g.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/bench_bf16_$g.json"))
print("bf16 cfg2 gather=$g", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
  ME_AMD_BF16_GATHER=$g timeout 300 python bench.py --workload conv4d --steps 20 --warmup 5 --cpu-budget 0 --dtype bf16 > $OUT/bench4d_bf16_$g.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/bench4d_bf16_$g.json"))
print("bf16 cfg5 gather=$g", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
  ME_AMD_BF16_GATHER=$g timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16_$g.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/unet_bf16_$g.json"))
print("bf16 unet gather=$g", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
ME_AMD_BF16_GATHER=1 timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --graph > $OUT/unet_bf16_graph.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/unet_bf16_graph.json"))
print("bf16 unet graph", d["value"], d["ms_per_step"])
PY
