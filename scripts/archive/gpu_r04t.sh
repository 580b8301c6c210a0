#!/bin/bash
# Round-4 session T: MinkUNet34C bf16 with and without the wave-specialised kernel; layer table; bf16 parity suite
set +e
OUT=$PWD/gpurun_out/r04t
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --pmc off > $OUT/unet_$name.json 2>$OUT/unet_$name.err
  python - <<PY
import json
d = json.loads(open("$OUT/unet_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"], {k: round(v["avg_ms"] * v.get("calls", 1), 3) if isinstance(v, dict) and "avg_ms" in v else v for k, v in d.get("kernels", {}).items()})
PY
}
run ws_off ME_AMD_BF16_WS=0
run ws_on A=1
run ws_on_depth2 ME_AMD_BF16_WS_DEPTH=2
run ws_off2 ME_AMD_BF16_WS=0
python scripts/unet_layers.py > $OUT/layers_ws.log 2>&1
ME_AMD_BF16_WS=0 python scripts/unet_layers.py > $OUT/layers_old.log 2>&1
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_minkunet.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.log
