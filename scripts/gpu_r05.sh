#!/bin/bash
# round-5 GPU sessions: scripts/gpu_r05.sh <step> [args]; everything lands in gpurun_out/r05_<step>*.log
set -u
mkdir -p gpurun_out
step=${1:-halo1}; shift || true
case "$step" in
  halo1)   # first contact of the halo kernel: parity tests, then the per-layer sweep
    timeout 900 python -m pytest tests/test_gpu_halo.py -x -q 2>&1 | tail -25 > gpurun_out/r05_halo1_pytest.log
    cat gpurun_out/r05_halo1_pytest.log
    timeout 600 python scripts/halo_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_halo1_sweep.log
    cat gpurun_out/r05_halo1_sweep.log
    ;;
  halo_prof)   # kernel durations of the sweep on one level (rocprofv3 --kernel-trace --stats)
    cd /tmp && export TMPDIR=/tmp
    LEVELS=${LEVELS:-8} REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_halo -o halo -- python $GRAFT_REPO_ROOT/scripts/halo_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/r05_halo_prof_sweep.log 2>&1
    cd $GRAFT_REPO_ROOT
    f=$(find /tmp/prof_halo -name "*kernel_stats.csv" | head -1)
    python3 - "$f" > gpurun_out/r05_halo_prof_stats.txt <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if len(r) >= 4 and ("me::" in r[0] or r[0] == "Name"):
        print(r[0][:78].ljust(78), r[1].rjust(6), r[3][:9].rjust(10))
PY
    head -30 gpurun_out/r05_halo_prof_stats.txt
    tail -8 gpurun_out/r05_halo_prof_sweep.log
    ;;
  halo_sweep)  # LEVELS / CONFIGS / TARGET from the environment
    timeout 900 python scripts/halo_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_halo_sweep_${TAG:-x}.log
    cat gpurun_out/r05_halo_sweep_${TAG:-x}.log
    ;;
  rccl)   # RCCL first contact + bench --gpus 1 --backend nccl (multi_gpu block) + the default bench line
    timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_halo.py -x -q 2>&1 | tail -15 > gpurun_out/r05_rccl_pytest.log
    cat gpurun_out/r05_rccl_pytest.log
    timeout 600 python bench.py --gpus 1 --backend nccl --steps 20 --warmup 5 --cpu-budget 0 --pmc off > gpurun_out/r05_bench_nccl1.json 2> gpurun_out/r05_bench_nccl1.err
    tail -3 gpurun_out/r05_bench_nccl1.err
    python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05_bench_nccl1.json").read().strip().splitlines()[-1])
print("headline", l["value"], l["ms_per_step"], "multi_gpu", json.dumps(l.get("multi_gpu"))[:400])
for k, w in (l.get("workloads") or {}).items():
    print(k, {x: w.get(x) for x in ("value", "ms_per_step", "error")}, json.dumps(w.get("multi_gpu"))[:300], json.dumps(w.get("timing")), json.dumps(w.get("gpu_state")))
PY
    ;;
  bench)   # the driver's command
    timeout 900 python bench.py > gpurun_out/r05_bench_${TAG:-default}.json 2> gpurun_out/r05_bench_${TAG:-default}.err
    tail -3 gpurun_out/r05_bench_${TAG:-default}.err
    python - <<PY
import json
l = json.loads(open("gpurun_out/r05_bench_${TAG:-default}.json").read().strip().splitlines()[-1])
print("headline", l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"].get("traffic"))
for k, w in (l.get("workloads") or {}).items():
    print(k, {x: w.get(x) for x in ("value", "ms_per_step", "error")}, json.dumps(w.get("hip_graph")), json.dumps(w.get("timing")), json.dumps(w.get("gpu_state")))
PY
    ;;
  stem)    # stacked-offset kernel: parity tests, the bf16 layer tests it now serves, sweep against the tile-plan kernel
    timeout 600 python -m pytest tests/test_gpu_stem.py tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -25 | tee gpurun_out/r05_stem_tests.log
    grep -q " passed" gpurun_out/r05_stem_tests.log && ! grep -q "failed\|error\|core" gpurun_out/r05_stem_tests.log || exit 1
    timeout 300 python scripts/stem_sweep.py 2>&1 | grep -v "^$" | tee gpurun_out/r05_stem_sweep.txt
    (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stem -o stem -- python $GRAFT_REPO_ROOT/scripts/stem_sweep.py > /dev/null 2>&1; python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/prof_stem/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(t in r['Name'] for t in ('stem', 'conv_tile', 'conv_ws', 'k_conv')):
            print(r['Name'][:90], r['Calls'], 'avg_us', float(r['AverageNs']) / 1e3, 'min_us', float(r['MinNs']) / 1e3)
PY
    ) 2>&1 | tee gpurun_out/r05_stem_kernels.txt
    timeout 300 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 --pmc off > gpurun_out/r05_stem_unet.json 2> gpurun_out/r05_stem_unet.err; tail -c 1500 gpurun_out/r05_stem_unet.json
    ;;
  suite)   # the whole GPU suite + smoke
    timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_pytest_gpu_${TAG:-x}.log
    cat gpurun_out/r05_pytest_gpu_${TAG:-x}.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee gpurun_out/r05_smoke_${TAG:-x}.log
    ;;
  final)   # closing session: GPU suite, smoke, the driver's bench command, kernel statistics of the default line and of the MinkUNet34C step
    OUT=gpurun_out/r05_final; mkdir -p $OUT
    timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee $OUT/smoke.log
    timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json | head -c 300; echo
    cd /tmp && export TMPDIR=/tmp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_def -o default -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-budget 0 --extra-workloads off --pmc off > $GRAFT_REPO_ROOT/$OUT/prof_default.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_unet -o unet -- python $GRAFT_REPO_ROOT/bench.py --workload minkunet --dtype bf16 --steps 5 --warmup 3 --cpu-budget 0 --pmc off --min-time 0 --min-blocks 5 --max-blocks 5 --no-gpu-state > $GRAFT_REPO_ROOT/$OUT/prof_unet.log 2>&1
    cd $GRAFT_REPO_ROOT
    cp $(find /tmp/prof_def -name "*kernel_stats.csv" | head -1) $OUT/rocprof_kernel_stats_default.csv
    cp $(find /tmp/prof_unet -name "*kernel_stats.csv" | head -1) $OUT/rocprof_kernel_stats_minkunet34c_bf16.csv
    head -8 $OUT/rocprof_kernel_stats_default.csv | cut -c1-160
    timeout 600 python scripts/unet_layers.py > $OUT/layers_minkunet34c_bf16.log 2>&1; head -12 $OUT/layers_minkunet34c_bf16.log
    # a new scene every step: lazily, as one recipe replay, with a loader thread
    for mode in "fresh" "fresh --replay-maps" "pipelined"; do
      timeout 300 python bench.py --workload minkunet --dtype bf16 --scenes $mode --steps 10 --warmup 3 --cpu-budget 0 --pmc off --no-gpu-state --no-graph-probe 2> /dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('scenes $mode', d.get('ms_per_step'), 'ms/step', d.get('timing', {}).get('blocks_ms_per_step'))" | tee -a $OUT/unet_scenes.log
    done
    ;;
  *) echo "unknown step $step"; exit 2;;
esac
