"""Per-kernel register / scratch / LDS / occupancy table of every gfx950 kernel in csrc/*.hip, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (compile-time facts: runs without a GPU).
usage: python scripts/kernel_resources.py [-o out.txt]   (compiles each .hip once, ~4 min)"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "minkowskiengine_amd", "csrc")
FIELDS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill",
          "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def main():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("-o", "--output", default=None, help="write the table to this file instead of stdout")
    args = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            log = os.path.join(tmp, os.path.basename(src) + ".log")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I",
                   os.path.join(ROOT, "include"), "-Wno-unused-function", "-Wno-unused-lambda-capture",
                   "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(tmp, "x.o." + os.path.basename(src))]
            procs.append((src, log, subprocess.Popen(cmd, stdout=open(log, "w"), stderr=subprocess.STDOUT)))
        for src, log, p in procs:
            p.wait()
            cur = None
            for line in open(log):
                m = re.search(r"remark:\s+Function Name: (\S+)", line)
                if m:
                    cur = {"file": os.path.basename(src), "name": m.group(1)}
                    rows.append(cur)
                    continue
                m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
                if m and cur is not None:
                    cur[m.group(1).strip()] = m.group(2)
    names = demangle([r["name"] for r in rows])
    lines = ["# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (scripts/kernel_resources.py)",
             "# file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spill | SGPR spill | waves/SIMD | static LDS B"]
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        lines.append(" | ".join([r["file"], n] + [r.get(f, "?") for f in FIELDS]))
    text = "\n".join(lines) + "\n"
    if args.output:
        open(args.output, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
