#!/bin/bash
# Does the GPU hold its 2.4 GHz clock under the conv kernel?  (1) MFMA-rate micro-benchmark, (2) sclk / power
# samples from sysfs while the forward kernel runs back to back, (3) GRBM_GUI_ACTIVE cycles per launch.
set +e
TAG=${1:-clock}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate scripts/ubench/mfma_rate.hip 2>/dev/null
timeout 120 /tmp/mfma_rate 3000 2>&1 | tee $OUT/mfma_rate.log
echo "== sysfs sensors"
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $H"; ls $H 2>/dev/null | tr '\n' ' '; echo
sample() { for i in $(seq 1 $1); do
    echo "t=$i sclk=$(cat $H/freq1_input 2>/dev/null) mclk=$(cat $H/freq2_input 2>/dev/null) power_uW=$(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null) temp=$(cat $H/temp1_input 2>/dev/null)"; sleep 0.25; done; }
echo "idle:"; sample 3 | tee $OUT/idle.log
for dt in f32 bf16; do
  echo "== forward kernel back to back ($dt)"
  DTYPE=$dt BWD=0 ITERS=30000 timeout 200 python scripts/prof_conv.py > $OUT/loop_$dt.log 2>&1 &
  PID=$!
  sleep 6            # import torch + map build
  sample 12 | tee $OUT/busy_$dt.log
  wait $PID
  tail -1 $OUT/loop_$dt.log
done
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|mclk" | head -8
echo "== GRBM cycles per launch"
cd /tmp
BWD=0 ITERS=20 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/grbm -o grbm -- python $REPO/scripts/prof_conv.py > $OUT/grbm.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections
out = "$OUT"
dur = collections.defaultdict(list)
for f in glob.glob(out + "/grbm/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:50]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/grbm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in cnt.items():
    if "conv_tile" not in k and "wgrad" not in k: continue
    d = sum(dur[k]) / max(len(dur[k]), 1)
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, f"mean duration {d/1e3:.1f} us (with counters on)",
          {c: f"{sum(v)/len(v)/d:.3f} cycles/ns" for c, v in cs.items()})
PY
