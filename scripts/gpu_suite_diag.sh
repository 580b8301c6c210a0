#!/bin/bash
# the GPU suite with everything captured + the box's identity (a run of the suite aborted on one box of the pool: this
# prints what a next occurrence needs)
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/suite_diag_box.log
import torch, os, subprocess
p = torch.cuda.get_device_properties(0)
print("device", p.name, "CUs", p.multi_processor_count, "mem GiB", round(p.total_memory / 2**30, 1), "gcn", getattr(p, "gcnArchName", "?"))
print("free/total", [round(v / 2**30, 1) for v in torch.cuda.mem_get_info()])
print(subprocess.run("rocm-smi --showuniqueid --showmemuse --showcomputepartition --showmemorypartition 2>/dev/null | grep -v '^=' | head -20", shell=True, capture_output=True, text=True).stdout)
PY
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/suite_diag.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/suite_diag.log | tail -1
L=$(grep -n "Fatal Python" gpurun_out/suite_diag.log | head -1 | cut -d: -f1)
if [ -n "$L" ]; then s=$((L-15)); [ $s -lt 1 ] && s=1; sed -n "${s},$((L+6))p" gpurun_out/suite_diag.log | cut -c1-300; fi
