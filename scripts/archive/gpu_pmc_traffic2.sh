#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE in separate passes) + TCC hit rate of the forward tile kernels:
# round-1 kernel and LDS-DMA kernel, row tiles and spatial tiles.
set +e
TAG=${1:-r02_traffic}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $REPO/scripts/prof_conv.py > $OUT/$n.log 2>&1
  echo "$n rc=$?"; }
export BWD=1 ITERS=5
for cfg in "v8rows 0 0 0 rows" "v8spatial 0 0 0 spatial" "dmarows 3100 196 4 rows" "dmaspatial 3100 196 4 spatial"; do
  set -- $cfg
  export VARIANT=$2 TILE=$3 CAP=$4 ME_AMD_TILE_ORDER=$5
  run ${1}_f FETCH_SIZE
  run ${1}_w WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
done
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:58]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", os.path.basename(d.rstrip("/")))
        for k, cs in agg.items():
            if "conv_tile" not in k and "wgrad_f32" not in k: continue
            print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
find $OUT -name "*.csv" -size +2M -delete
echo done
