#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r04j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "spatial or tile" 2>&1 | tail -3 | tee $OUT/pytest_tiles.log
timeout 600 python scripts/tile_dispatch_sweep.py 2>&1 | grep -v amdgpu | tee $OUT/tile_dispatch.log
