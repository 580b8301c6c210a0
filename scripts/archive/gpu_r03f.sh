#!/bin/bash
# Round-3 session F: the native host layer — its tests, then host time (enqueue) and step time of MinkUNet34C bf16 on the
# native vs the python host, eager and as a hipGraph; the headline on both hosts.
set +e
OUT=$PWD/gpurun_out/r03f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_native_host.py tests/test_reference_package.py -m gpu -q --timeout 600 > $OUT/pytest_native.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_native.log | tail -12
for h in native python; do
  ME_AMD_HOST=$h timeout 300 python scripts/host_cprofile_unet.py > $OUT/host_cprofile_$h.log 2>&1; echo "== $h"; head -8 $OUT/host_cprofile_$h.log | grep -v amdgpu
  ME_AMD_HOST=$h timeout 300 python scripts/host_profile2.py > $OUT/host_profile2_$h.log 2>&1; grep -E "forward \+ backward|module call" $OUT/host_profile2_$h.log
  ME_AMD_HOST=$h timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16_$h.json 2>$OUT/unet_bf16_$h.err
  ME_AMD_HOST=$h timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --graph > $OUT/unet_bf16_graph_$h.json 2>$OUT/unet_bf16_graph_$h.err
  ME_AMD_HOST=$h timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --scenes fresh > $OUT/unet_bf16_fresh_$h.json 2>$OUT/unet_bf16_fresh_$h.err
  ME_AMD_HOST=$h timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --scenes pipelined > $OUT/unet_bf16_pipelined_$h.json 2>$OUT/unet_bf16_pipelined_$h.err
  ME_AMD_HOST=$h timeout 300 python bench.py --cpu-budget 0 --extra-workloads off > $OUT/bench_$h.json 2>$OUT/bench_$h.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["ms_per_step"], d["timing"]["blocks_ms_per_step"][:3])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
tail -5 $OUT/unet_bf16_native.err
