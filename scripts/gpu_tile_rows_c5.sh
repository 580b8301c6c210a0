#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r02_tiles_c5; mkdir -p $OUT
for dt in f32 bf16; do
for T in 0 256 224 196 176 160 131; do
  ME_AMD_TILE_ROWS=$T timeout 300 python bench.py --workload conv4d --dtype $dt --steps 20 --warmup 5 --cpu-budget 0 > $OUT/c5_${dt}_T$T.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/c5_${dt}_T$T.json"))
print("cfg5 $dt T=$T", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
done
done
