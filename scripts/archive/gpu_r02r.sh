#!/bin/bash
set +e
TAG=${1:-r02r}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -20
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $REPO/scripts/prof_conv.py > $OUT/$n.log 2>&1
  echo "$n rc=$?"; }
export BWD=1 ITERS=5 VARIANT=0 TILE=0 CAP=0
run x3_f FETCH_SIZE
run x3_w WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/x3_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:70]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", os.path.basename(d.rstrip("/")))
        for k, cs in agg.items():
            if "conv_tile" not in k and "wgrad_f32" not in k: continue
            print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench.json 2>$OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("f32 cfg2", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["roofline"]["frac"], d["roofline"].get("frac_of_pipe"))
PY
for dt in f32 bf16; do
timeout 600 python bench.py --workload minkunet --dtype $dt --steps 10 --warmup 3 --cpu-budget 0 > $OUT/unet_$dt.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/unet_$dt.json"))
print("$dt unet", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
timeout 300 python bench.py --workload conv4d --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench4d.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench4d.json"))
print("f32 cfg5", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
find $OUT -name "*.csv" -size +2M -delete
