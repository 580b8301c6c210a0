#!/bin/bash
# Round-4 session D: the offset-synchronous bf16 kernel — bit-identity tests, then the schedule sweep over the MinkUNet levels.
set +e
OUT=$PWD/gpurun_out/r04d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -x -k "offset_synchronous or split_k" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
timeout 900 python scripts/offsync_sweep.py > $OUT/offsync_sweep.log 2>&1; grep -v amdgpu $OUT/offsync_sweep.log
