#!/bin/bash
# Spatial tile order (needs the position-space maps: ME_AMD_SPATIAL_MAPS=1) with the final kernels: config 2,
# MinkUNet34C steps (cached / fresh scenes), PMC traffic on config 2
set +e
OUT=$PWD/gpurun_out/r02_exp9
mkdir -p $OUT
export ME_AMD_SPATIAL_MAPS=1 ME_AMD_TILE_ORDER=spatial
B="--cpu-budget 0 --workload minkunet --steps 10 --warmup 3"
timeout 300 python bench.py --cpu-budget 0 > $OUT/c2_f32_spatial.json 2>/dev/null
timeout 300 python bench.py --cpu-budget 0 --dtype bf16 > $OUT/c2_bf16_spatial.json 2>/dev/null
timeout 300 python bench.py --cpu-budget 0 --extent 215 > $OUT/c2_sparse_spatial.json 2>/dev/null
timeout 300 python bench.py $B --dtype bf16 > $OUT/unet_bf16_spatial.json 2>/dev/null
timeout 300 python bench.py $B --dtype f32 > $OUT/unet_f32_spatial.json 2>/dev/null
timeout 300 python bench.py $B --dtype bf16 --scenes fresh > $OUT/unet_bf16_fresh_spatial.json 2>/dev/null
timeout 300 python bench.py $B --dtype bf16 --scenes pipelined > $OUT/unet_bf16_pipelined_spatial.json 2>/dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = {n: round(v.get("ms_per_step", 0), 4) for n, v in d.get("kernels", {}).items()}
    print(os.path.basename(f), d["value"], d["ms_per_step"], "ms", k, d.get("cold_ms"))
PY
bash scripts/gpu_pmc_final.sh r02_pmc_spatial
