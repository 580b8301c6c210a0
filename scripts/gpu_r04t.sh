#!/bin/bash
# the tests that need the tuning build (-DME_DEBUG_VARIANTS), on a tagged library next to the default one
set +e
OUT=$PWD/gpurun_out/r04t
mkdir -p $OUT
ME_AMD_HOST=python ME_AMD_LIB_TAG=dbg timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bf16.py tests/test_abi_symbols.py -m gpu -q --timeout 900 > $OUT/pytest_debug_build.log 2>&1
grep -v amdgpu.ids $OUT/pytest_debug_build.log | tail -3
