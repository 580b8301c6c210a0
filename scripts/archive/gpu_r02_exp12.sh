#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r02_exp12
mkdir -p $OUT
timeout 200 python scripts/host_gap_profile.py > $OUT/rows.log 2>&1
ME_AMD_SPATIAL_TILES=1 timeout 200 python scripts/host_gap_profile.py > $OUT/zorder.log 2>&1
EXTENT=215 DTYPE=f32 timeout 200 python scripts/host_gap_profile.py > $OUT/sparse_rows.log 2>&1
EXTENT=215 DTYPE=f32 ME_AMD_SPATIAL_TILES=1 timeout 200 python scripts/host_gap_profile.py > $OUT/sparse_zorder.log 2>&1
for f in rows zorder sparse_rows sparse_zorder; do echo "== $f"; grep -v "^$" $OUT/$f.log | grep -v amdgpu.ids | head -34 | cut -c1-150; done
