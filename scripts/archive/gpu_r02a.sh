#!/bin/bash
# Round 2, first GPU session: smoke, the GPU parity suite (per-element tolerances, bf16 pooling, config 3, streams,
# 2 ranks on 1 GPU), bench lines (N = 1, N = 2 oversubscribed, minkunet eager / graph).
set +e
TAG=${1:-r02a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(rocm-smi --showproductname; nproc; free -g) > $OUT/env.log 2>&1
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -2 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
if grep -q "failed" $OUT/pytest_gpu.log; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu_all.log 2>&1
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_all.log | head -40
fi
cat gpurun_out/config3_parity.log 2>/dev/null
echo "== bench N=1"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== bench N=2 (self-spawned, oversubscribed -> gloo)"
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_n2.json 2> $OUT/bench_n2.err; echo "bench n2 rc=$?"
cut -c1-900 $OUT/bench_n2.json; tail -5 $OUT/bench_n2.err
echo "== minkunet bf16 eager / graph, f32"
timeout 900 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 30 > $OUT/bench_unet_bf16.json 2> $OUT/bench_unet_bf16.err; echo "rc=$?"
cut -c1-700 $OUT/bench_unet_bf16.json; tail -3 $OUT/bench_unet_bf16.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/%s/bench_unet_bf16.json" % "r02a"))
    print("cpu_baseline:", d.get("cpu_baseline"))
except Exception as e:
    print("no unet line", e)
PY
timeout 600 python bench.py --workload minkunet --dtype bf16 --steps 10 --warmup 3 --cpu-budget 0 --graph > $OUT/bench_unet_bf16_graph.json 2> $OUT/bench_unet_bf16_graph.err; echo "graph rc=$?"
cut -c1-400 $OUT/bench_unet_bf16_graph.json; tail -5 $OUT/bench_unet_bf16_graph.err
timeout 600 python bench.py --workload minkunet --dtype bf16 --gpus 2 --steps 5 --warmup 2 --cpu-budget 0 > $OUT/bench_unet_n2.json 2> $OUT/bench_unet_n2.err; echo "unet n2 rc=$?"
cut -c1-500 $OUT/bench_unet_n2.json; tail -5 $OUT/bench_unet_n2.err
timeout 300 python examples/multigpu_ddp.py --oversubscribe 2 --iterations 3 --points 50000 > $OUT/ddp_example.log 2>&1; echo "ddp example rc=$?"
tail -5 $OUT/ddp_example.log
echo "== done"
