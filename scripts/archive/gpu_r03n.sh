#!/bin/bash
# A/B of the bf16 tile kernel's operand-read blocking (ME_BF16_KSB) and dual accumulator chains (ME_BF16_TWO):
# per-layer table of a MinkUNet34C bf16 step for each tagged library build.
set +e
OUT=$PWD/gpurun_out/r03n
mkdir -p $OUT
export ME_AMD_HOST=python
for tag in "" k1 k2 k4n; do
  ME_AMD_LIB_TAG=$tag timeout 300 python scripts/unet_layers.py > $OUT/layers_${tag:-default}.log 2>&1
  head -1 $OUT/layers_${tag:-default}.log
done
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_minkunet.py -m gpu -q -x --timeout 600 > $OUT/pytest_bf16.log 2>&1
tail -3 $OUT/pytest_bf16.log
