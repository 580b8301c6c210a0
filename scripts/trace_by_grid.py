"""Average duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV: separates the shapes of one kernel."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
if not rows:
    sys.exit("empty trace")
k = rows[0].keys()
name = [c for c in k if c.lower() in ("kernel_name", "name")][0]
gs = [c for c in k if c.lower() in ("grid_size", "grid_size_x", "workgroup_count")]
s0 = [c for c in k if c.lower().startswith("start")][0]
e0 = [c for c in k if c.lower().startswith("end")][0]
acc = collections.OrderedDict()
for r in rows:
    key = (re.sub(r"\(.*", "", r[name])[:60], tuple(r[c] for c in gs if c in r))
    acc.setdefault(key, []).append((int(r[e0]) - int(r[s0])) / 1e3)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (n, g), v in acc.items():
    if flt and flt not in n: continue
    v = sorted(v)
    print(f"{n:60s} grid {','.join(g):>14s} calls {len(v):5d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
