#!/bin/bash
set +e
TAG=${1:-r02c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SPECS="3064:196:3,3100:196:4,3100:196:4,3064:196:3,0:0:0,3100:196:4" SHAPES="70:64:128" timeout 600 python scripts/tune_conv_dma.py > $OUT/tune_dma.log 2>&1; echo "tune rc=$?"
cat $OUT/tune_dma.log | grep -v amdgpu.ids
echo "== done"
