#!/bin/bash
# Round-3 closing session: full GPU test suite, smoke(), the default bench line (with workloads + cpu_baseline) and its
# rocprofv3 kernel statistics, the other workloads' lines, the MinkUNet34C bf16 step's kernel statistics, the 2-rank line.
set +e
TAG=${1:-r03_final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --dtype bf16 --cpu-budget 0 --extra-workloads off > $OUT/bench_bf16.json 2>/dev/null
timeout 600 python bench.py --extent 215 --cpu-budget 0 --extra-workloads off > $OUT/bench_sparse.json 2>/dev/null
timeout 600 python bench.py --workload conv4d --cpu-budget 0 > $OUT/bench_conv4d.json 2>/dev/null
timeout 600 python bench.py --workload conv4d --dtype bf16 --cpu-budget 0 > $OUT/bench_conv4d_bf16.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --cpu-budget 0 --extra-workloads off > $OUT/bench_n2.json 2> $OUT/bench_n2.err; echo "bench n2 rc=$?"
timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 20 --warmup 5 --cpu-budget 0 > $OUT/unet_bf16.json 2>/dev/null
timeout 900 python bench.py --workload minkunet --dtype f32 --steps 10 --warmup 3 --cpu-budget 0 --no-graph-probe > $OUT/unet_f32.json 2>/dev/null
timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --no-graph-probe --scenes fresh > $OUT/unet_bf16_fresh.json 2>/dev/null
timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --no-graph-probe --scenes pipelined > $OUT/unet_bf16_pipelined.json 2>/dev/null
ME_AMD_HOST=python timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --no-graph-probe > $OUT/unet_bf16_python_host.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o trace -- python $REPO/bench.py --cpu-budget 0 --extra-workloads off > $OUT/prof_bench.json 2> $OUT/prof_bench.log
find $OUT/prof_bench -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_bench.csv \;
rm -rf $OUT/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_unet -o trace -- python $REPO/bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --no-graph-probe > $OUT/prof_unet.json 2> $OUT/prof_unet.log
find $OUT/prof_unet -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_unet_bf16.csv \;
rm -rf $OUT/prof_unet
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
          "frac", d["roofline"].get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d["config"].get("host_layer"))
PY
head -8 $OUT/kernel_stats_bench.csv | cut -c1-160
