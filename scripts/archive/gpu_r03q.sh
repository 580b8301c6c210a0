#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03q
mkdir -p $OUT
ME_AMD_HOST=python timeout 300 python scripts/bn_bandwidth.py > $OUT/bn_bandwidth.log 2>&1
grep -v amdgpu.ids $OUT/bn_bandwidth.log
