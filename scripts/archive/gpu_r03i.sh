#!/bin/bash
# Round-3 session I: multi-offset batches in the fp32-MFMA tile kernel (me_conv_target_f32_fused) — tests, config 5,
# sparse config 2, MinkUNet34C fp32, with and without; bf16 shape policy check.
set +e
OUT=$PWD/gpurun_out/r03i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bf16.py -m gpu -q --timeout 600 -x -k "multi_offset or slab_width or vs_oracle" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for f in 1 0; do
  ME_AMD_F32_FUSE=$f timeout 300 python bench.py --workload conv4d --cpu-budget 0 > $OUT/conv4d_fuse$f.json 2>/dev/null
  ME_AMD_F32_FUSE=$f timeout 300 python bench.py --extent 215 --cpu-budget 0 --extra-workloads off > $OUT/sparse_fuse$f.json 2>/dev/null
  ME_AMD_F32_FUSE=$f timeout 300 python bench.py --extent 215 --cin 32 --cout 32 --cpu-budget 0 --extra-workloads off > $OUT/sparse3232_fuse$f.json 2>/dev/null
  ME_AMD_F32_FUSE=$f timeout 600 python bench.py --workload minkunet --dtype f32 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_f32_fuse$f.json 2>/dev/null
done
ME_AMD_F32_FUSE=1 ME_AMD_F32_SPLIT=0 timeout 300 python bench.py --extent 215 --cpu-budget 0 --extra-workloads off > $OUT/sparse_nosplit_fuse1.json 2>/dev/null
ME_AMD_F32_FUSE=0 ME_AMD_F32_SPLIT=0 timeout 300 python bench.py --extent 215 --cpu-budget 0 --extra-workloads off > $OUT/sparse_nosplit_fuse0.json 2>/dev/null
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_bf16.json 2>/dev/null
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"].get("frac"), d.get("hip_graph", {}).get("ms_per_step"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
