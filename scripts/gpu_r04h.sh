#!/bin/bash
# coarse-level layers (5k / 21k voxels, 256 channels): fewer weight bytes per row with taller tiles on narrower slabs?
set +e
OUT=$PWD/gpurun_out/r04h
mkdir -p $OUT
run() { tag=$1; shift; env ME_AMD_HOST=python "$@" timeout 300 python scripts/unet_layers.py > $OUT/layers_$tag.log 2>&1; echo "$tag $(grep '^step' $OUT/layers_$tag.log)"; }
run default
run nc64_T78 BF16_SHAPE=64,128 ME_AMD_TILE_ROWS=78
run nc64_T156 BF16_SHAPE=64,128 ME_AMD_TILE_ROWS=156
run nc128_T78 ME_AMD_TILE_ROWS=78
run nc64_Tauto BF16_SHAPE=64,128
