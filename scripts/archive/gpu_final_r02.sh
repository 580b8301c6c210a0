#!/bin/bash
# Round-2 closing session: full GPU test suite, smoke(), the default bench line (with cpu_baseline) + its rocprofv3
# kernel statistics, the self-spawned 2-rank line, the other workloads' lines.
set +e
TAG=${1:-r02_final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --dtype bf16 --cpu-budget 0 > $OUT/bench_bf16.json 2>/dev/null
timeout 600 python bench.py --extent 215 --cpu-budget 0 > $OUT/bench_sparse.json 2>/dev/null
timeout 600 python bench.py --workload conv4d > $OUT/bench_conv4d.json 2>/dev/null
timeout 600 python bench.py --workload conv4d --dtype bf16 --cpu-budget 0 > $OUT/bench_conv4d_bf16.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --cpu-budget 0 > $OUT/bench_n2.json 2> $OUT/bench_n2.err; echo "bench n2 rc=$?"
timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 > $OUT/unet_bf16.json 2>/dev/null
timeout 900 python bench.py --workload minkunet --dtype f32 --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_f32.json 2>/dev/null
timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --graph > $OUT/unet_bf16_graph.json 2>/dev/null
timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --scenes fresh > $OUT/unet_bf16_fresh.json 2>/dev/null
timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --scenes pipelined > $OUT/unet_bf16_pipelined.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o trace -- python $REPO/bench.py --cpu-budget 0 > $OUT/prof_bench.json 2> $OUT/prof_bench.log
find $OUT/prof_bench -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_bench.csv \;
find $OUT/prof_bench -type f ! -name "*stats*" -size +1M -delete
cd $REPO
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = {n: round(v.get("avg_ms", v.get("ms_per_step", 0)), 4) for n, v in d.get("kernels", {}).items()}
    print(os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], "ms", "n_gpus", d["n_gpus"], k,
          "frac", d["roofline"].get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
head -8 $OUT/kernel_stats_bench.csv | cut -c1-160
