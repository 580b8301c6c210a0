#!/bin/bash
# Round-3 session A: GPU tests (incl. the reference package on the HIP kernels), the new default bench line with the
# `workloads` object, per-layer table of the bf16 MinkUNet34C step under tile-height / fusion / order settings, and the
# A/B of split3's non-finite handling on the headline kernels.
set +e
OUT=$PWD/gpurun_out/r03a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|skipped" $OUT/pytest_gpu.log | tail -8
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
ME_AMD_LIB_TAG=nonf timeout 300 python bench.py --cpu-budget 0 --extra-workloads off > $OUT/bench_no_nonfinite.json 2>/dev/null
timeout 300 python bench.py --cpu-budget 0 --extra-workloads off > $OUT/bench_nonfinite.json 2>/dev/null
for cfg in "base" "T32:ME_AMD_TILE_ROWS=32" "T64:ME_AMD_TILE_ROWS=64" "T128:ME_AMD_TILE_ROWS=128" "T256:ME_AMD_TILE_ROWS=256" \
           "fuse1:ME_AMD_BF16_FUSE=1" "fuse0:ME_AMD_BF16_FUSE=0" "spatial:ME_AMD_TILE_ORDER=spatial" "gather:ME_AMD_BF16_GATHER=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}; [ "$envs" = "$cfg" ] && envs=""
  env $envs timeout 300 python scripts/unet_layers.py > $OUT/layers_$name.log 2>&1
  echo "== $name: $(head -1 $OUT/layers_$name.log)"
done
python - <<PY
import json
for f in ("bench", "bench_no_nonfinite", "bench_nonfinite"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"].get("frac"),
          d.get("cold_breakdown_ms", {}).get("first_backward_plans"))
    for k, v in (d.get("workloads") or {}).items():
        print("  ", k, v.get("value"), v.get("ms_per_step"), v.get("wall_s"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
PY
