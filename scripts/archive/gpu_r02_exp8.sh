#!/bin/bash
# Policy experiments on the final kernels (no code change): tile order with the wave-specialised kernel on config 2,
# the split (bf16-pipe) kernels below the c_src*c_dst >= 8192 threshold on config 5 and on MinkUNet34C fp32.
set +e
OUT=$PWD/gpurun_out/r02_exp8
mkdir -p $OUT
B="--cpu-budget 0"
timeout 300 python bench.py $B > $OUT/c2_rows.json 2>/dev/null
ME_AMD_TILE_ORDER=spatial timeout 300 python bench.py $B > $OUT/c2_spatial.json 2>/dev/null
timeout 300 python bench.py $B --workload conv4d > $OUT/c5_auto.json 2>/dev/null
ME_AMD_F32_SPLIT=1 timeout 300 python bench.py $B --workload conv4d > $OUT/c5_split.json 2>/dev/null
ME_AMD_TILE_ORDER=rows timeout 300 python bench.py $B --workload conv4d > $OUT/c5_rows.json 2>/dev/null
ME_AMD_F32_SPLIT=1 timeout 400 python bench.py $B --workload minkunet --dtype f32 --steps 10 --warmup 3 > $OUT/unet_f32_split.json 2>/dev/null
timeout 400 python bench.py $B --workload minkunet --dtype f32 --steps 10 --warmup 3 > $OUT/unet_f32_auto.json 2>/dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = {n: round(v.get("avg_ms", v.get("ms_per_step", 0)), 4) for n, v in d.get("kernels", {}).items()}
    print(os.path.basename(f), d["value"], d["ms_per_step"], "ms", k, "frac", d["roofline"].get("frac"), d["roofline"].get("kernel"))
PY
