#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03s
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
ME_AMD_HOST=python timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o bn -- python $REPO/scripts/bn_bandwidth.py > $OUT/bn_bandwidth.log 2>&1
cd $REPO
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python scripts/trace_by_grid.py $f k_bn > $OUT/bn_by_grid.log 2>&1
cat $OUT/bn_by_grid.log
rm -rf $OUT/prof
