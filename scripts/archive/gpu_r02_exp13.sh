#!/bin/bash
# The matrix-bound fp32 kernels on spatially compact tiles by default (flat-table maps: spatial index of the target map)
set +e
OUT=$PWD/gpurun_out/r02_exp13
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_prefetch.py -q -x --timeout 600 > $OUT/pytest_conv.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_conv.log
B="--cpu-budget 0 --workload minkunet --steps 10 --warmup 3"
timeout 300 python bench.py --cpu-budget 0 > $OUT/c2_f32.json 2>/dev/null
timeout 300 python bench.py --cpu-budget 0 --extent 215 > $OUT/c2_sparse.json 2>/dev/null
timeout 300 python bench.py $B --dtype f32 > $OUT/unet_f32.json 2>/dev/null
timeout 300 python bench.py $B --dtype f32 --scenes fresh > $OUT/unet_f32_fresh.json 2>/dev/null
ME_AMD_TILE_ORDER=rows timeout 300 python bench.py $B --dtype f32 --scenes fresh > $OUT/unet_f32_fresh_rows.json 2>/dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = {n: round(v.get("ms_per_step", 0), 4) for n, v in d.get("kernels", {}).items()}
    print(os.path.basename(f), d["value"], d["ms_per_step"], "ms", k, d.get("cold_ms"), d.get("cold"))
PY
bash scripts/gpu_pmc_final.sh r02_pmc_tile_order_final
