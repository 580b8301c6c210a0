#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r04f
mkdir -p $OUT
ME_AMD_HOST=python ME_AMD_LIB_TAG=tim timeout 300 python scripts/bf16_phase_timing.py > $OUT/phase.log 2>&1
grep -v amdgpu.ids $OUT/phase.log
