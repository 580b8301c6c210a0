#!/bin/bash
set +e
TAG=${1:-r02_kmap}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for wl in conv3d conv4d; do
  for sp in 1 0; do
    WORKLOAD=$wl ME_AMD_SPATIAL_MAPS=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${wl}_$sp -o t -- python $REPO/scripts/prof_kmap.py > $OUT/${wl}_$sp.log 2>&1
    echo "== $wl spatial=$sp"
    python - <<PY
import csv, glob
f = glob.glob("$OUT/${wl}_$sp/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total us per rep", tot / 5e3)
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/5e3:8.1f} us/rep calls/rep {int(r['Calls'])/5:5.1f} avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:80]}")
PY
  done
done
find $OUT -name "*.csv" -size +2M -delete
