#!/bin/bash
# HBM-side traffic PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss) for the config-2 kernels, fp32 and bf16.
# usage: gpurun -- 'bash scripts/gpu_pmc_traffic.sh tag'
set +e
TAG=${1:-pmc_traffic}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $REPO/scripts/prof_conv.py > $OUT/$n.log 2>&1
  echo "$n rc=$?"; }
run f_fetch FETCH_SIZE
run f_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
DTYPE=bf16 run b_fetch FETCH_SIZE
DTYPE=bf16 run b_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:70]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", os.path.basename(d.rstrip("/")))
        for k, cs in agg.items():
            if "conv_tile" not in k and "wgrad" not in k: continue
            print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
