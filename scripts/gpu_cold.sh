#!/bin/bash
# cold-vs-warm MinkUNet34C step: wall time and rocprofv3 kernel stats of both
set +e
TAG=${1:-cold}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for m in warm cold; do MODE=$m timeout 300 python scripts/cold_profile.py 2>&1 | tail -1 | tee -a $OUT/wall.log; done
cd /tmp
for m in warm cold; do
  MODE=$m timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$m -o trace -- python $OLDPWD/scripts/cold_profile.py > $OUT/prof_$m.log 2>&1
  tail -1 $OUT/prof_$m.log
  find $OUT/prof_$m -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_$m.csv \;
  find $OUT/prof_$m -type f ! -name "*stats*" -size +2M -delete
done
cd $OLDPWD
python - <<P
import csv
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r['Name']]=(int(r['Calls']),float(r['TotalDurationNs']))
    return d
w=load('$OUT/kernel_stats_warm.csv'); c=load('$OUT/kernel_stats_cold.csv')
steps=13.0
rows=[]
for k,(n,t) in c.items():
    wn,wt=w.get(k,(0,0.0))
    rows.append(((t-wt)/steps/1e3,k,n/steps,wn/steps))
rows.sort(reverse=True)
print("extra GPU us/step   calls/step cold (warm)   kernel")
tot=0
for d,k,n,wn in rows[:40]:
    print(f"{d:10.1f}  {n:7.1f} ({wn:6.1f})  {k[:110]}")
for d,*_ in rows: tot+=d
print(f"total extra GPU time {tot/1e3:.2f} ms/step; GPU total warm {sum(t for _,t in w.values())/steps/1e6:.2f} cold {sum(t for _,t in c.values())/steps/1e6:.2f} ms/step")
P
