#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r04i
mkdir -p $OUT
export TMPDIR=/tmp
for dt in f32 bf16; do DTYPE=$dt timeout 300 python scripts/wgrad_locality_sweep.py 2>&1 | grep -v amdgpu | tee -a $OUT/wgrad_locality.log; done
