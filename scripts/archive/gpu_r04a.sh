#!/bin/bash
# Round-4 session A: L2 -> CU streaming micro-benchmark, the full GPU suite (now on both host layers), per-layer baseline.
set +e
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 scripts/ubench/l2_stream > $OUT/l2_stream.log 2>&1; echo "ubench rc=$?"; cat $OUT/l2_stream.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
ME_AMD_HOST=python timeout 300 python scripts/unet_layers.py > $OUT/layers_base.log 2>&1; head -40 $OUT/layers_base.log
