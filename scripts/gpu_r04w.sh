#!/bin/bash
# Round-4 session W: batch fusion in the wave-specialised bf16 kernel — parity, the step, the layer table
set +e
OUT=$PWD/gpurun_out/r04w
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "wave_specialised or oracle" 2>&1 | tail -4 | tee $OUT/pytest_ws.log
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --pmc off > $OUT/unet_$name.json 2>$OUT/unet_$name.err
  python - <<PY
import json
d = json.loads(open("$OUT/unet_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"])
PY
}
cat > /tmp/nofuse.py <<'PY'
PY
run ws_fuse A=1
ME_AMD_HOST=python python scripts/unet_layers.py > $OUT/layers_ws_fuse.log 2>&1
head -3 $OUT/layers_ws_fuse.log
