#!/bin/bash
set +e
TAG=${1:-r02h}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "ME_AMD_SPATIAL_MAPS=0" "ME_AMD_SPATIAL_MAPS=1 ME_AMD_TILE_ORDER=rows" "ME_AMD_SPATIAL_MAPS=1 ME_AMD_TILE_ORDER=spatial"; do
  echo "== $cfg"
  env $cfg timeout 300 python scripts/cold_minkunet.py 2>&1 | tail -1
  env $cfg DTYPE=f32 timeout 300 python scripts/cold_minkunet.py 2>&1 | tail -1
done
