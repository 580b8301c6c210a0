#!/bin/bash
# Round-3 session K: PMC counters of the bf16 tile / weight-gradient kernels on two MinkUNet34C layer shapes (what are
# the waves waiting for?)
set +e
OUT=$PWD/gpurun_out/r03k
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
run() { n=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $REPO/scripts/prof_layer.py > $OUT/$n.log 2>&1
  echo "$n rc=$?"; }
for shape in "ts8:LEVEL=8 CIN=256 COUT=256" "ts1:LEVEL=1 CIN=96 COUT=96" "ts16:LEVEL=16 CIN=256 COUT=256"; do
  tag=${shape%%:*}; export ${shape#*:}
  run ${tag}_p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
  run ${tag}_p2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU
  run ${tag}_p3 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
  run ${tag}_p4 FETCH_SIZE GRBM_GUI_ACTIVE
done
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:48]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", os.path.basename(d.rstrip("/")))
        for k, cs in agg.items():
            if "conv_tile" not in k and "wgrad" not in k: continue
            print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        dur = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            dur[row["Kernel_Name"][:48]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            if "conv_tile" in k or "wgrad" in k: print("   us:", k, round(sum(v) / len(v), 1), "n=", len(v))
PY
