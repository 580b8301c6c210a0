#!/bin/bash
set +e
TAG=${1:-r02s}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for mode in "0 rows" "1 rows" "1 spatial"; do
  set -- $mode
  ME_AMD_SPATIAL_MAPS=$1 ME_AMD_TILE_ORDER=$2 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_$1_$2.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/bench_$1_$2.json"))
print("f32 cfg2 spatial_maps=$1 tiles=$2", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
done
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $REPO/scripts/prof_conv.py > $OUT/$n.log 2>&1
  echo "$n rc=$?"; }
export BWD=1 ITERS=5 VARIANT=0 TILE=0 CAP=0 ME_AMD_SPATIAL_MAPS=1 ME_AMD_TILE_ORDER=spatial
run x3sp_f FETCH_SIZE
run x3sp_w WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/x3sp_*/")):
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
find $OUT -name "*.csv" -size +2M -delete
