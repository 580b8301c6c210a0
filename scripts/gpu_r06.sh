#!/bin/bash
# round-6 GPU sessions: scripts/gpu_r06.sh <step> [args]; everything lands in gpurun_out/r06_<step>*.log
set -u
mkdir -p gpurun_out
step=${1:-base}; shift || true
summ() {  # headline summary of a bench JSON line
python - "$1" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = l["roofline"]
print(sys.argv[1].split("/")[-1], "ms/step", l["ms_per_step"], "Mpts/s", round(l["value"], 1), "frac", r.get("frac"),
      "traffic_by_pass", json.dumps(r.get("traffic_by_pass")), "kernels_us", json.dumps(l.get("kernel_us") or l.get("timers") or {})[:300])
for k, w in (l.get("workloads") or {}).items():
    print("   ", k, {x: w.get(x) for x in ("value", "ms_per_step", "error")})
PY
}
case "$step" in
  base)   # state of the tree on the GPU: whole GPU suite + the driver's bench command
    timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension\|amdgpu.ids" | tail -15 > gpurun_out/r06_base_pytest.log
    cat gpurun_out/r06_base_pytest.log
    timeout 900 python bench.py > gpurun_out/r06_bench_base.json 2> gpurun_out/r06_bench_base.err
    tail -3 gpurun_out/r06_bench_base.err
    summ gpurun_out/r06_bench_base.json
    ;;
  headline_knobs)   # headline only, with the existing locality knobs (traffic per pass from the in-run PMC passes)
    for cfg in "default:" "spatialmaps:ME_AMD_SPATIAL_MAPS=1" "dispatch:ME_AMD_TILE_DISPATCH=1" "both:ME_AMD_SPATIAL_MAPS=1 ME_AMD_TILE_DISPATCH=1"; do
      tag=${cfg%%:*}; envs=${cfg#*:}
      env $envs timeout 600 python bench.py --extra-workloads off --cpu-budget 0 --pmc on > gpurun_out/r06_knob_$tag.json 2> gpurun_out/r06_knob_$tag.err
      tail -2 gpurun_out/r06_knob_$tag.err
      summ gpurun_out/r06_knob_$tag.json
    done
    ;;
  rowwise)   # row-wise kernel: parity tests (+ the bf16 layer tests it now serves), per-layer table of the MinkUNet34C step per variant
    timeout 1200 python -m pytest tests/test_gpu_rowwise.py tests/test_gpu_bf16.py tests/test_gpu_minkunet.py tests/test_gpu_norm.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension\|amdgpu.ids" | tail -25 | tee gpurun_out/r06_rowwise_tests.log
    for cfg in "on:ME_AMD_ROWWISE=1" "g1:ME_AMD_ROWWISE=1 ME_AMD_RW_G=1" "g2:ME_AMD_ROWWISE=1 ME_AMD_RW_G=2" "off:ME_AMD_ROWWISE=0"; do
      tag=${cfg%%:*}; envs=${cfg#*:}
      env $envs timeout 600 python scripts/unet_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_layers_rowwise_$tag.log
      echo "== $tag"; head -1 gpurun_out/r06_layers_rowwise_$tag.log
      grep -E "\((1|8), " gpurun_out/r06_layers_rowwise_$tag.log | grep -v wgrad | sort -k6 | awk '{printf "%s %s %s%s%s%s%s%s | ", $3, $5, $6,$7,$8,$9,$10,$11} END {print ""}'
      env $envs timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --pmc off --extra-workloads off > gpurun_out/r06_unet_rowwise_$tag.json 2> gpurun_out/r06_unet_rowwise_$tag.err
      python -c "import json; l=json.loads(open('gpurun_out/r06_unet_rowwise_$tag.json').read().strip().splitlines()[-1]); print('unet $tag', l['ms_per_step'], l['value'])"
    done
    ;;
  insert)   # fused insert: equivalence tests + cold numbers of the bench (insert_ms / insert_GBs) + the rowwise / norm tests again
    timeout 1500 python -m pytest tests/test_gpu_coords.py tests/test_gpu_rowwise.py tests/test_gpu_norm.py tests/test_gpu_prefetch.py tests/test_gpu_native_host.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension\|amdgpu.ids" | tail -15 | tee gpurun_out/r06_insert_tests.log
    timeout 600 python bench.py --extra-workloads off --cpu-budget 0 --pmc off > gpurun_out/r06_bench_insert.json 2> gpurun_out/r06_bench_insert.err
    python -c "import json; l=json.loads(open('gpurun_out/r06_bench_insert.json').read().strip().splitlines()[-1]); print('headline', l['ms_per_step'], 'cold', json.dumps(l['cold']))"
    for sc in cached fresh pipelined; do
      timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --pmc off --extra-workloads off --scenes $sc $( [ $sc = fresh ] && echo --replay-maps ) > gpurun_out/r06_unet_scenes_$sc.json 2> gpurun_out/r06_unet_scenes_$sc.err
      python -c "import json; l=json.loads(open('gpurun_out/r06_unet_scenes_$sc.json').read().strip().splitlines()[-1]); print('unet $sc', l['ms_per_step'], l['value'])"
    done
    ;;
  insert_prof)   # kernel statistics of insert_and_map (fused vs the scan pipeline)
    for f in 1 0; do
      (cd /tmp && export TMPDIR=/tmp && ME_INSERT_FUSED=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ins$f -o ins -- python $GRAFT_REPO_ROOT/scripts/insert_prof.py 2>&1 | grep insert_and_map)
      python3 - /tmp/prof_ins$f <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.reader(open(f)):
        if len(r) >= 4 and r[0] != "Name":
            print(r[0][:80].ljust(80), r[1].rjust(6), f"{float(r[3])/1e3:9.2f} us")
PY
    done
    ;;
  rowwise_bench)   # microbenchmark + ablations of the row-wise kernel
    BASE=1 python scripts/rowwise_bench.py 2>&1 | grep -v amdgpu.ids
    # (the ablations of round 6 — no stores 25.2 -> 16.4 us, no loads behind the first ring 20.6, no MFMAs 23.5 — needed a
    # kernel argument that is not in the tree any more: profiles/r06_rowwise_bench.log)
    for g in 1 2; do ME_AMD_RW_G=$g python scripts/rowwise_bench.py 2>&1 | grep -v amdgpu.ids; done
    SHAPE=200000,96,96 python scripts/rowwise_bench.py 2>&1 | grep -v amdgpu.ids
    SHAPE=80000,192,128 python scripts/rowwise_bench.py 2>&1 | grep -v amdgpu.ids
    ;;
  fresh_prof)   # kernel statistics of the MinkUNet34C step with a new scene every step (recipe replay) and cached
    for sc in fresh cached; do
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$sc -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload minkunet --dtype bf16 --scenes $sc $( [ $sc = fresh ] && echo --replay-maps ) --steps 16 --warmup 4 --min-time 0 --max-blocks 1 --cpu-budget 0 --pmc off --extra-workloads off --no-graph-probe --no-gpu-state > $GRAFT_REPO_ROOT/gpurun_out/r06_prof_bench_$sc.json 2> /dev/null)
      find /tmp/prof_$sc -name "*kernel_stats*.csv" -exec cp {} gpurun_out/r06_rocprof_kernel_stats_minkunet34c_bf16_$sc.csv \;
      python -c "import json; print('$sc', json.loads(open('gpurun_out/r06_prof_bench_$sc.json').read().strip().splitlines()[-1])['ms_per_step'])"
    done
    ;;
  sq2)   # LDS / VALU / VMEM activity counters of the headline kernels (fractions of the wave cycles)
    python scripts/sq_pass2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_sq_pass2_${WL:-conv3d}_${DT:-f32}.log
    ;;
  sq)   # the counter-based MFMA utilisation pass of bench.py (headline + MinkUNet34C entry)
    timeout 900 python bench.py --cpu-budget 0 --pmc on > gpurun_out/r06_bench_sq.json 2> gpurun_out/r06_bench_sq.err
    tail -3 gpurun_out/r06_bench_sq.err
    python - <<'PY'
import json
l = json.loads(open("gpurun_out/r06_bench_sq.json").read().strip().splitlines()[-1])
r = l["roofline"]
print("headline", l["ms_per_step"], "frac", r.get("frac"), "mfma_busy_frac", r.get("mfma_busy_frac"))
for k, v in (r.get("mfma_busy_by_pass") or {}).items():
    print("  ", k, json.dumps(v))
w = (l.get("workloads") or {}).get("minkunet34c_bf16_200k") or {}
rr = w.get("roofline") or {}
print("minkunet", w.get("ms_per_step"), "mfma_busy_frac", rr.get("mfma_busy_frac"), "traffic", rr.get("traffic"))
for k, v in (rr.get("mfma_busy_by_kernel") or {}).items():
    print("  ", k[:70], json.dumps(v))
PY
    ;;
  final)   # closing session: GPU suite, smoke, the driver's bench command, kernel statistics of the default line and of the MinkUNet34C step
    OUT=gpurun_out/r06_final; mkdir -p $OUT
    timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension\|amdgpu.ids" | tail -8 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee $OUT/smoke.log
    timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; summ $OUT/bench_default.json
    (cd /tmp && export TMPDIR=/tmp
     timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_def -o default -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-budget 0 --extra-workloads off --pmc off > $GRAFT_REPO_ROOT/$OUT/prof_default.log 2>&1
     timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_unet -o unet -- python $GRAFT_REPO_ROOT/bench.py --workload minkunet --dtype bf16 --steps 5 --warmup 3 --cpu-budget 0 --pmc off --min-time 0 --min-blocks 5 --max-blocks 5 --no-gpu-state --extra-workloads off > $GRAFT_REPO_ROOT/$OUT/prof_unet.log 2>&1)
    cp $(find /tmp/prof_def -name "*kernel_stats.csv" | head -1) $OUT/rocprof_kernel_stats_default.csv
    cp $(find /tmp/prof_unet -name "*kernel_stats.csv" | head -1) $OUT/rocprof_kernel_stats_minkunet34c_bf16_cached.csv
    head -8 $OUT/rocprof_kernel_stats_default.csv | cut -c1-160
    timeout 600 python scripts/unet_layers.py 2>&1 | grep -v amdgpu.ids > $OUT/layers_minkunet34c_bf16.log; head -12 $OUT/layers_minkunet34c_bf16.log
    for mode in "cached" "fresh" "fresh --replay-maps" "pipelined"; do
      timeout 300 python bench.py --workload minkunet --dtype bf16 --scenes $mode --steps 10 --warmup 3 --cpu-budget 0 --pmc off --no-gpu-state --no-graph-probe --extra-workloads off 2> /dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('scenes $mode', d.get('ms_per_step'), 'ms/step', d.get('timing', {}).get('blocks_ms_per_step'))" | tee -a $OUT/unet_scenes.log
    done
    ;;
  *) echo "unknown step $step"; exit 1 ;;
esac
