#!/bin/bash
set +e
TAG=${1:-r02e}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_coords.py tests/test_gpu_conv.py tests/test_gpu_pooling.py -m gpu -q --timeout 600 -x > $OUT/pytest_coords.log 2>&1; echo "tests rc=$?"
tail -5 $OUT/pytest_coords.log
bash scripts/gpu_prof_kmap.sh $TAG 2>&1 | grep -v "^\s*[0-4]\.[0-9] us"
