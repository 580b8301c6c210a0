#!/bin/bash
# Round-4 session C: float64 tests, split-K / occupancy sweep (+ rocprofv3 kernel trace of the 5k-voxel level),
# bench.py --gpus 2 on the one GPU (gloo): the multi_gpu block and the MinkUNet34C DDP entry.
set +e
OUT=$PWD/gpurun_out/r04c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_f64.py -m gpu -q --timeout 300 > $OUT/pytest_f64.log 2>&1; echo "pytest f64 rc=$?"; tail -15 $OUT/pytest_f64.log
timeout 600 python scripts/splitk_sweep.py > $OUT/splitk_sweep.log 2>&1; grep -v amdgpu $OUT/splitk_sweep.log
cd /tmp
LEVELS=16 CONFIGS="0,0,0,0;0,0,2,0;0,0,4,0;128,128,2,1" REPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_splitk -o splitk -- python $GRAFT_REPO_ROOT/scripts/splitk_sweep.py > $OUT/prof_splitk.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r04c/prof_splitk/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "k_conv_tile_bf16" in r["Name"] or "splitk" in r["Name"]:
            print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_n2.json 2> $OUT/bench_n2.err; echo "bench n2 rc=$?"; tail -3 $OUT/bench_n2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04c/bench_n2.json").read().strip().splitlines()[-1])
print("n2", d["value"], d["ms_per_step"], json.dumps(d["multi_gpu"])[:900])
w = d.get("workloads", {})
for k, v in w.items():
    print(k, v.get("value"), v.get("ms_per_step"), json.dumps(v.get("multi_gpu"))[:900] if isinstance(v, dict) else v)
PY
