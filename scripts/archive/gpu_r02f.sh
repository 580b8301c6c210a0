#!/bin/bash
set +e
TAG=${1:-r02f}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -30
for sp in 1 0; do
  ME_AMD_SPATIAL_MAPS=$sp timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_sp$sp.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/bench_sp$sp.json"))
print("conv3d spatial=$sp", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["cold"]["kernel_map_ms"], d["cold"]["plans_ms"], d["cold"]["insert_ms"])
PY
done
ME_AMD_SPATIAL_MAPS=1 timeout 300 python bench.py --workload conv4d --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_conv4d.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench_conv4d.json"))
print("conv4d", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["cold"])
PY
SPECS="0:0:0,0:0:0,3100:196:4,3064:196:3" SHAPES="70:64:128" timeout 300 python scripts/tune_conv_dma.py 2>&1 | grep -v amdgpu
ME_AMD_SPATIAL_MAPS=0 SPECS="0:0:0,0:0:0,3100:196:4,3064:196:3" SHAPES="70:64:128" timeout 300 python scripts/tune_conv_dma.py 2>&1 | grep -v amdgpu
bash scripts/gpu_prof_kmap.sh ${TAG}_kmap 2>&1 | grep "==\|total\|probe\|k_sp\|k_rs\|scan_single\|tile_order" 
