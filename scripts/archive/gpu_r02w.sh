#!/bin/bash
set +e
TAG=${1:-r02w}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -k "fusion or vs_oracle" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -20
for v in 1 0; do
  ME_AMD_BF16_FUSE=$v timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16_v$v.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/unet_bf16_v$v.json"))
print("bf16 unet variant $v", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
  ME_AMD_BF16_FUSE=$v timeout 300 python bench.py --workload conv4d --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 > $OUT/c4d_bf16_v$v.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/c4d_bf16_v$v.json"))
print("bf16 cfg5 variant $v", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
  ME_AMD_BF16_FUSE=$v timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 > $OUT/c2_bf16_v$v.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/c2_bf16_v$v.json"))
print("bf16 cfg2 variant $v", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
  ME_AMD_BF16_FUSE=$v timeout 300 python bench.py --dtype bf16 --extent 215 --steps 20 --warmup 5 --cpu-budget 0 > $OUT/c2s_bf16_v$v.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/c2s_bf16_v$v.json"))
print("bf16 cfg2 sparse variant $v", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
done
