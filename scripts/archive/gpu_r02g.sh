#!/bin/bash
set +e
TAG=${1:-r02g}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_$name.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/bench_$name.json"))
print("$name", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, "kmap", d["cold"]["kernel_map_ms"], "plans", d["cold"]["plans_ms"])
PY
}
run v8_rows ME_AMD_TILE_ORDER=rows
run v8_spatial ME_AMD_TILE_ORDER=spatial
run dma_rows ME_AMD_TILE_ORDER=rows ME_AMD_CONV_VARIANT=3100 ME_AMD_TILE_ROWS=196 ME_AMD_BATCH_GROUPS=4
run dma_spatial ME_AMD_TILE_ORDER=spatial ME_AMD_CONV_VARIANT=3100 ME_AMD_TILE_ROWS=196 ME_AMD_BATCH_GROUPS=4
run v8_rows_again ME_AMD_TILE_ORDER=rows

for to in rows spatial; do
  ME_AMD_TILE_ORDER=$to timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --dtype bf16 > $OUT/bench_bf16_$to.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/bench_bf16_$to.json"))
print("bf16 $to", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
done
