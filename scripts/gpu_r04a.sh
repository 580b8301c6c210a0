#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
ME_AMD_HOST=python BF16_SHAPE=128,0 timeout 300 python scripts/unet_layers.py > $OUT/layers_nc128.log 2>&1
grep "^step" $OUT/layers_nc128.log
ME_AMD_HOST=python timeout 300 python scripts/unet_layers.py > $OUT/layers_policy.log 2>&1
grep "^step" $OUT/layers_policy.log
