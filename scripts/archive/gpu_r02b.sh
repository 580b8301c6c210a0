#!/bin/bash
set +e
TAG=${1:-r02b}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/config3_parity.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -30
cat gpurun_out/config3_parity.log 2>/dev/null
echo "== tune dma"
timeout 600 python scripts/tune_conv_dma.py > $OUT/tune_dma.log 2>&1; echo "tune rc=$?"
cat $OUT/tune_dma.log | grep -v amdgpu.ids
echo "== done"
