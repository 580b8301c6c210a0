#!/bin/bash
# Round-4: the tests of the tuning-build variants (skipped on the default library) on libme_amd_dbg.so
# (ME_AMD_LIB_TAG=dbg ME_AMD_EXTRA_HIPCC_FLAGS=-DME_DEBUG_VARIANTS python -m minkowskiengine_amd.build); Python host layer
# (the native module links the default library).
set +e
OUT=$PWD/gpurun_out/r04_dbg
mkdir -p $OUT
export TMPDIR=/tmp ME_AMD_LIB_TAG=dbg ME_AMD_HOST=python
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_conv.py -q -m gpu --timeout 900 \
  -k "offset_synchronous or batch_fusion_is_bit_identical or weight_gradient_is_bit_identical or sparse_fp32_launches or lds_dma or wave_specialised" \
  > $OUT/pytest_debug_build.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_debug_build.log
