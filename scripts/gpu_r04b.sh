#!/bin/bash
# full GPU suite + smoke + default bench line (with the workloads object)
set +e
OUT=$PWD/gpurun_out/r04b
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1
grep -v amdgpu.ids $OUT/pytest_gpu.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1
tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r04b/bench.json').read().strip().split('\n')[-1])
    print('headline', d['value'], d['ms_per_step'], d['config'].get('host_layer'))
    for w,e in d.get('workloads',{}).items(): print('  ', w, e.get('value'), e.get('ms_per_step'))
except Exception as e: print('unreadable', e)
PY
