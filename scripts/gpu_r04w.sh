#!/bin/bash
set +e
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_norm.py -m gpu -q --timeout 300 2>&1 | tail -1
timeout 300 python bench.py --cpu-budget 0 --extra-workloads off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"
