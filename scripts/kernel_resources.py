"""Compact register / spill / occupancy table of one csrc/*.hip translation unit (hipcc -Rpass-analysis=kernel-resource-usage):
    python scripts/kernel_resources.py conv_halo [substring of the demangled kernel name]"""
import os
import re
import subprocess
import sys

unit = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("ME_AMD_EXTRA_HIPCC_FLAGS", "").split() + \
      ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(root, "minkowskiengine_amd", "csrc", unit + ".hip"), "-o", f"/tmp/{unit}.res.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name.split("(")[0]}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(\S[^:]*): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1)] = m.group(2)
for r in rows:
    if pat in r["name"]:
        print("%-72s vgpr %s agpr %s sgpr-spill %s vgpr-spill %s scratch %s occ %s" % (
            r["name"][:72], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
            r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
