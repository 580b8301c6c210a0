#!/bin/bash
# Round-3 session B: side-stream weight gradient, one-launch weight packing, 128-column / 256-channel bf16 tiles,
# channel padding, cheaper non-finite handling — tests first, then A/B of each on MinkUNet34C bf16 and the headline.
set +e
OUT=$PWD/gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pack.py tests/test_gpu_conv.py tests/test_gpu_bf16.py tests/test_gpu_minkunet.py tests/test_gpu_distributed.py tests/test_gpu_prefetch.py -m gpu -q --timeout 600 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | tail -8
run() { name=$1; shift; env "$@" timeout 300 python scripts/unet_layers.py > $OUT/layers_$name.log 2>&1; echo "== $name: $(grep '^step' $OUT/layers_$name.log)"; }
run new
run old ME_AMD_WGRAD_STREAM=0 ME_AMD_PACK_CACHE=0 ME_AMD_PAD_CHANNELS=0 BF16_SHAPE=64,0
run no_stream ME_AMD_WGRAD_STREAM=0
run no_packcache ME_AMD_PACK_CACHE=0
run no_pad ME_AMD_PAD_CHANNELS=0
run nc64 BF16_SHAPE=64,0
run nc128_kc128 BF16_SHAPE=128,128
run nc0_kc128 BF16_SHAPE=0,128
for v in "new" "nostream:ME_AMD_WGRAD_STREAM=0" "nopack:ME_AMD_PACK_CACHE=0" "nonf:ME_AMD_LIB_TAG=nonf"; do
  name=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs="X=1"
  env $envs timeout 300 python bench.py --cpu-budget 0 --extra-workloads off > $OUT/bench_$name.json 2>/dev/null
done
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16.json 2>/dev/null
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --graph > $OUT/unet_bf16_graph.json 2>$OUT/unet_bf16_graph.err
timeout 600 python bench.py --workload minkunet --dtype f32 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_f32.json 2>/dev/null
timeout 600 python bench.py --workload conv4d --cpu-budget 0 > $OUT/conv4d.json 2>/dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"].get("frac"))
PY
