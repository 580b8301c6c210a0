#!/bin/bash
# Round-4 session K: MinkUNet34C bf16 step by tile dispatch (heaviest first / XCD chunks) and the source-size rule for
# spatial tiles; per-layer table of the adopted setting.
set +e
OUT=$PWD/gpurun_out/r04k
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --pmc off > $OUT/unet_$name.json 2>$OUT/unet_$name.err
  python - <<PY
import json
d = json.loads(open("$OUT/unet_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"], sorted(d["timing"]["blocks_ms_per_step"])[:4] if "timing" in d else "")
PY
}
run base ME_AMD_TILE_DISPATCH=0 ME_AMD_TILE_SPATIAL_SRC_MB=0
run spatial28 ME_AMD_TILE_DISPATCH=0 ME_AMD_TILE_SPATIAL_SRC_MB=28
run xcd ME_AMD_TILE_DISPATCH=1 ME_AMD_TILE_SPATIAL_SRC_MB=0
run xcd_spatial28 ME_AMD_TILE_DISPATCH=1 ME_AMD_TILE_SPATIAL_SRC_MB=28
run spatial18 ME_AMD_TILE_DISPATCH=0 ME_AMD_TILE_SPATIAL_SRC_MB=18
run base2 ME_AMD_TILE_DISPATCH=0 ME_AMD_TILE_SPATIAL_SRC_MB=0
