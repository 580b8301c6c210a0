#!/bin/bash
# Round-4 session H: what the per-launch HIP events cost the headline line; tile height of the headline.
set +e
OUT=$PWD/gpurun_out/r04h
mkdir -p $OUT
export TMPDIR=/tmp
for tb in 1 4 1000 1 1000; do
  python bench.py --steps 20 --warmup 5 --cpu-budget 0 --extra-workloads off --pmc off --timer-blocks $tb > $OUT/b_$tb.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$OUT/b_$tb.json").read().strip().splitlines()[-1])
print("timer-blocks $tb", d["value"], d["ms_per_step"], d["timing"]["fastest_block_ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
done
for T in 160 192 256; do
  ME_AMD_TILE_ROWS=$T python bench.py --steps 20 --warmup 5 --cpu-budget 0 --extra-workloads off --pmc off > $OUT/t_$T.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$OUT/t_$T.json").read().strip().splitlines()[-1])
print("tile rows $T", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
done
