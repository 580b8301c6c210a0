#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r02_exp14
mkdir -p $OUT
for i in 1 2; do
  ME_AMD_TILE_ORDER=rows timeout 300 python bench.py --cpu-budget 0 --extent 215 > $OUT/sparse_rows_$i.json 2>/dev/null
  timeout 300 python bench.py --cpu-budget 0 --extent 215 > $OUT/sparse_auto_$i.json 2>/dev/null
done
EXTENT=215 DTYPE=f32 ME_AMD_TILE_ORDER=rows TOP=1 timeout 200 python scripts/host_gap_profile.py 2>&1 | grep "host enqueue"
EXTENT=215 DTYPE=f32 TOP=1 timeout 200 python scripts/host_gap_profile.py 2>&1 | grep "host enqueue"
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = {n: round(v.get("ms_per_step", 0), 4) for n, v in d.get("kernels", {}).items()}
    print(os.path.basename(f), d["value"], d["ms_per_step"], "ms", k, d["timing"]["blocks_ms_per_step"][:6])
PY
