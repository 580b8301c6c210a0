#!/bin/bash
# Round-4 session B: split-K of the bf16 tile kernel — tests, G sweep on the coarse levels, the MinkUNet34C step.
set +e
OUT=$PWD/gpurun_out/r04b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_pack.py tests/test_gpu_norm.py tests/test_gpu_native_host.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 600 python scripts/splitk_sweep.py > $OUT/splitk_sweep.log 2>&1; cat $OUT/splitk_sweep.log | grep -v amdgpu
for g in 0 -1; do
  BF16_SPLITK=$g ME_AMD_HOST=python timeout 300 python scripts/unet_layers.py > $OUT/layers_splitk_$g.log 2>&1; head -12 $OUT/layers_splitk_$g.log | grep -v amdgpu
done
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("$OUT/unet_bf16.json").read().strip().splitlines()[-1])
print("unet_bf16", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()}, d.get("hip_graph"))
PY
