#!/bin/bash
set +e
TAG=${1:-r02_pmc_wgrad}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $REPO/scripts/prof_conv.py > $OUT/$n.log 2>&1
  echo "$n rc=$?"; }
export BWD=1 ITERS=5 VARIANT=0 TILE=0 CAP=0
for o in -1 0; do
  export WGRAD_ORDER=$o
  run order${o}_f FETCH_SIZE
  run order${o}_h TCC_HIT_sum TCC_MISS_sum
done
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/order*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:60]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        print("==", os.path.basename(d.rstrip("/")))
        for k, cs in agg.items():
            if "wgrad_f32" not in k: continue
            print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "avg_us", round(sum(dur[k]) / len(dur[k]) / 1e3, 1))
PY
find $OUT -name "*.csv" -size +2M -delete
