#!/bin/bash
set +e
TAG=${1:-r02l}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q --timeout 600 -s -k "not lds_dma" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|relative to" $OUT/pytest.log | head -40
for g in 1 0; do
  ME_AMD_F32_SPLIT=$g timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_$g.json 2>$OUT/bench_$g.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_$g.json"))
print("f32 cfg2 split=$g", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["frac"])
PY
done
for T in 96 128 160 176; do
  ME_AMD_TILE_ROWS=$T timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_T$T.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/bench_T$T.json"))
print("f32 cfg2 split T=$T", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
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
