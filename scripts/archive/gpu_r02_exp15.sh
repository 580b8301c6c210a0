#!/bin/bash
# Config 5 (sparse map): tile height sweep of the tile kernels — per-tile cost (weights, prologue) vs per-batch cost
set +e
OUT=$PWD/gpurun_out/r02_exp15
mkdir -p $OUT
for t in 64 128 256; do
  ME_AMD_TILE_ROWS=$t timeout 100 python bench.py --cpu-budget 0 --workload conv4d --dtype bf16 --min-blocks 2 > $OUT/bf16_t$t.json 2>/dev/null
  ME_AMD_TILE_ROWS=$t timeout 100 python bench.py --cpu-budget 0 --workload conv4d --min-blocks 2 > $OUT/f32_t$t.json 2>/dev/null
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = {n: round(v.get("ms_per_step", 0), 4) for n, v in d.get("kernels", {}).items()}
    print(os.path.basename(f), d["value"], d["ms_per_step"], "ms", k)
PY
