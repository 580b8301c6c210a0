#!/bin/bash
# timing ablations of the bf16 tile kernel (INVALID results): no LDS operand reads / accumulators written without the
# read / both — how much of a layer is LDS traffic
set +e
OUT=$PWD/gpurun_out/r04c
mkdir -p $OUT
for tag in "" noaread noacc noboth; do
  ME_AMD_HOST=python ME_AMD_LIB_TAG=$tag timeout 300 python scripts/unet_layers.py > $OUT/layers_${tag:-default}.log 2>&1
  grep "^step" $OUT/layers_${tag:-default}.log
done
