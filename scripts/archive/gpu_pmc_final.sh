#!/bin/bash
# HBM-side traffic of the round's default config-2 kernels (forward / dgrad: k_conv_tile_f32x3_ws, wgrad: k_wgrad_f32x3)
set +e
TAG=${1:-r02_pmc_final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $REPO/scripts/prof_conv.py > $OUT/$n.log 2>&1
  echo "$n rc=$?"; }
export BWD=1 ITERS=5 VARIANT=0 TILE=0 CAP=0
run final_f FETCH_SIZE
run final_w WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/final_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:60]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", os.path.basename(d.rstrip("/")))
        for k, cs in agg.items():
            if "conv_tile" not in k and "wgrad" not in k: continue
            print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
find $OUT -name "*.csv" -size +2M -delete
