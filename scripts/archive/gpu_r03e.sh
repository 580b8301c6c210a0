#!/bin/bash
set +e
OUT=$PWD/gpurun_out/r03e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pack.py tests/test_gpu_bf16.py tests/test_gpu_minkunet.py -m gpu -q --timeout 600 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python scripts/host_cprofile_unet.py > $OUT/host_cprofile.log 2>&1; head -8 $OUT/host_cprofile.log | grep -v amdgpu
for cfg in "graph:X=1" "graph_nostream:ME_AMD_WGRAD_STREAM=0" "graph_old:ME_AMD_WGRAD_STREAM=0 ME_AMD_PACK_CACHE=0 ME_AMD_PAD_CHANNELS=0 ME_AMD_FUSE_RESIDUAL=0" "graph_stream_all:ME_AMD_WGRAD_STREAM_MAX_ROWS=100000000" "graph_stream_100k:ME_AMD_WGRAD_STREAM_MAX_ROWS=100000" "graph_stream_25k:ME_AMD_WGRAD_STREAM_MAX_ROWS=25000" "graph_nopad:ME_AMD_PAD_CHANNELS=0" "graph_nofuse:ME_AMD_FUSE_RESIDUAL=0" "graph_nopack:ME_AMD_PACK_CACHE=0" "graph_nc64:X=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  extra=""; [ "$name" = "graph_nc64" ] && extra="--debug-bf16-shape 64,0"
  env $envs timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --graph $extra > $OUT/unet_bf16_$name.json 2>$OUT/unet_bf16_$name.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/unet_bf16_$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['timing']['blocks_ms_per_step'][:4])
except Exception as e: print('$name', 'failed', e)"
done
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16_eager.json 2>/dev/null
timeout 600 python bench.py --workload minkunet --dtype f32 --steps 10 --warmup 3 --cpu-budget 0 --graph > $OUT/unet_f32_graph.json 2>/dev/null
python - <<PY
import json
for n in ("unet_bf16_eager", "unet_f32_graph"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1]); print(n, d['ms_per_step'])
    except Exception as e: print(n, 'failed', e)
PY
