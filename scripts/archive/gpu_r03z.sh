#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03z
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $REPO/scripts/cold_trace.py > $OUT/cold.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_cold.csv \;
rm -rf $OUT/prof
cd $REPO
grep -v amdgpu.ids $OUT/cold.log | tail -3
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03z/kernel_stats_cold.csv')))
for r in rows[:40]: print('%-80s %5d %8.1f us  total %8.1f'%(r['Name'][:80], int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/15))
PY
timeout 600 python -m pytest tests/test_gpu_coords.py tests/test_gpu_conv.py -m gpu -q -x --timeout 600 2>&1 | tail -2
