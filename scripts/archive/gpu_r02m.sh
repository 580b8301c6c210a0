#!/bin/bash
set +e
TAG=${1:-r02m}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q --timeout 600 -k "vs_oracle or fp32_grade or fixtures or full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -40
for T in 0 ${TILES:-196 160 131 98}; do
  ME_AMD_TILE_ROWS=$T timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_T$T.json 2>$OUT/bench_T$T.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_T$T.json"))
print("f32 cfg2 split T=$T", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["frac"])
PY
done
ME_AMD_F32_SPLIT=1 timeout 300 python bench.py --workload conv4d --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench4d_1.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench4d_1.json"))
print("f32 cfg5 split=1", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
ME_AMD_F32_SPLIT=1 timeout 600 python bench.py --workload minkunet --dtype f32 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_f32_1.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/unet_f32_1.json"))
print("f32 unet split=1", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
