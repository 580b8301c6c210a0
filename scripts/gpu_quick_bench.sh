#!/bin/bash
# Short bench sweep (no CPU baseline): config 2 dense / sparse, config 5, MinkUNet34C; fp32 and bf16.
run() { python bench.py "$@" --cpu-budget 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '|', d['value'], 'Mpts/s', d['ms_per_step'], 'ms', {k:v['avg_ms'] for k,v in d['kernels'].items()})"; }
for dt in f32 bf16; do
  run --dtype $dt
  run --dtype $dt --extent 215
  run --dtype $dt --workload conv4d --steps 30 --warmup 5
  run --dtype $dt --workload minkunet --steps 10 --warmup 3
done
