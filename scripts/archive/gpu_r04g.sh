#!/bin/bash
# Round-4 session G: gather only the batch's own rows — bf16 tests, per-level sweep, MinkUNet34C step + per-layer table.
set +e
OUT=$PWD/gpurun_out/r04g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_minkunet.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
KNOB=twobuf MODES=0,1 timeout 600 python scripts/offsync_sweep.py > $OUT/sweep.log 2>&1; grep -v amdgpu $OUT/sweep.log
ME_AMD_HOST=python timeout 300 python scripts/unet_layers.py > $OUT/layers.log 2>&1; head -45 $OUT/layers.log | grep -v amdgpu
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --pmc off > $OUT/unet_bf16.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("$OUT/unet_bf16.json").read().strip().splitlines()[-1])
print("unet_bf16", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()}, d.get("hip_graph"))
PY
