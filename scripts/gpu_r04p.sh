#!/bin/bash
# two ranks sharing the one GPU: DDP MinkUNet34C bf16 line (with and without SyncBN) and the DDP example
set +e
OUT=$PWD/gpurun_out/r04p
mkdir -p $OUT
timeout 600 python bench.py --gpus 2 --workload minkunet --dtype bf16 --steps 5 --warmup 2 --cpu-budget 0 --no-graph-probe > $OUT/unet_n2.json 2> $OUT/unet_n2.err; echo "rc=$?"
timeout 600 python bench.py --gpus 2 --workload minkunet --dtype bf16 --steps 5 --warmup 2 --cpu-budget 0 --no-graph-probe --sync-bn --imbalance > $OUT/unet_n2_syncbn_imbalance.json 2> $OUT/unet_n2_syncbn.err; echo "rc=$?"
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], "ms n_gpus", d["n_gpus"], d["config"].get("parallelism"), d["config"].get("imbalance"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
tail -3 $OUT/unet_n2.err | cut -c1-200
