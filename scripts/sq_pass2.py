"""Second SQ counter pass over the headline kernels (LDS / VALU / VMEM activity): scripts/gpu_r06.sh sq2"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sets = [["SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
         "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS"],
        ["SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC",
         "SQ_INST_CYCLES_VMEM_RD", "SQ_VALU_MFMA_COEXEC_CYCLES"]]
wl, dt = os.environ.get("WL", "conv3d"), os.environ.get("DT", "f32")
for cs in sets:
    agg = bench.pmc_pass(wl, dt, cs, 3, 200)
    if not agg:
        print("pass failed", cs)
        continue
    for name, c in agg.items():
        if not any(m in name for m in bench.PMC_CONV_KERNELS):
            continue
        per = {k: v[0] / max(v[1], 1) for k, v in c.items()}
        wc = per.get("SQ_WAVE_CYCLES", 0) or 1
        short = name.split("(")[0].replace("void ", "").replace("me::", "")[:60]
        print(short, json.dumps({k: (round(v / wc, 4) if k.startswith("SQ_") and k != "SQ_WAVE_CYCLES" else round(v, 1)) for k, v in per.items()}))
