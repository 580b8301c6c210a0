#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03w
mkdir -p $OUT
ME_AMD_HOST=python timeout 300 python scripts/unet_layers.py > $OUT/layers.log 2>&1
grep -v amdgpu.ids $OUT/layers.log | grep "step\|wgrad" | head -40
