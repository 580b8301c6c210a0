#!/bin/bash
# timing ablations of the bf16 tile kernel (INVALID results), second set + the debug-build tests
set +e
OUT=$PWD/gpurun_out/r04d
mkdir -p $OUT
for tag in "" a1 a2 a3 a4 a5 a6; do
  ME_AMD_HOST=python ME_AMD_LIB_TAG=$tag timeout 300 python scripts/unet_layers.py > $OUT/layers_${tag:-default}.log 2>&1
  echo "$tag $(grep '^step' $OUT/layers_${tag:-default}.log)"
done
ME_AMD_HOST=python ME_AMD_LIB_TAG=dbg timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bf16.py tests/test_abi_symbols.py -m gpu -q --timeout 900 > $OUT/pytest_debug_build.log 2>&1
grep -v amdgpu.ids $OUT/pytest_debug_build.log | tail -3
