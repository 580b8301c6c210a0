#!/bin/bash
set +e
TAG=${1:-r02o}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bf16.py -m gpu -q --timeout 600 -k "not lds_dma" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -40
python scripts/phase_timing_x3.py > $OUT/phase.log 2>&1; cat $OUT/phase.log | grep -v amdgpu.ids
for T in 0 ${TILES}; do
  ME_AMD_TILE_ROWS=$T timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_T$T.json 2>$OUT/bench_T$T.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_T$T.json"))
print("f32 cfg2 split T=$T", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["frac"])
PY
done
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --dtype bf16 > $OUT/bench_bf16.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench_bf16.json"))
print("bf16 cfg2", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
timeout 300 python bench.py --workload conv4d --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench4d_1.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench4d_1.json"))
print("f32 cfg5 split=1", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
for dt in f32 bf16; do
timeout 600 python bench.py --workload minkunet --dtype $dt --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_$dt.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/unet_$dt.json"))
print("$dt unet", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
