#!/bin/bash
# PMC passes for the conv kernels.  usage: bash scripts/gpu_pmc.sh tag
set +e
TAG=${1:-pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -oE "\b(SQ|TA|TCP|TCC|TD|GRBM)_[A-Z0-9_]+" $OUT/counters.txt | sort -u > $OUT/counter_names.txt
wc -l $OUT/counter_names.txt
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $REPO/scripts/prof_conv.py > $OUT/$n.log 2>&1
  echo "$n rc=$?"
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
run p2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU
run p3 TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run p4 FETCH_SIZE
run p5 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run p6 GRBM_GUI_ACTIVE GRBM_COUNT
VARIANT=16 run d1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
VARIANT=16 run d3 TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
cd $REPO
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:60]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", os.path.basename(d.rstrip("/")))
        for k, cs in agg.items():
            if "conv" not in k and "wgrad" not in k: continue
            print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
find $OUT -name "*.csv" -size +3M -delete
