#!/bin/bash
set +e
TAG=${1:-r02ws}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q --timeout 600 -k "wave_specialised or vs_oracle or fixtures or full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head
for v in 0 31; do
  ME_AMD_CONV_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_v$v.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/bench_v$v.json"))
print("f32 cfg2 variant $v", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["frac"])
PY
  ME_AMD_CONV_VARIANT=$v timeout 600 python bench.py --workload minkunet --dtype f32 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_f32_v$v.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/unet_f32_v$v.json"))
print("f32 unet variant $v", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
