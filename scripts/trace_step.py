"""Ordered kernel timeline of the last full training step in a rocprofv3 kernel trace (between the last two
adam_kernel launches): index, start offset, duration, grid, short kernel name — the raw material for the per-tier /
per-layer tables under profiles/.

    python scripts/trace_step.py <dir with *_kernel_trace.csv> [--summary]
"""
import csv
import glob
import re
import sys

d = sys.argv[1]
fn = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[adam[-2] + 1:adam[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])


def short(k):
    k = k.replace("void ", "").replace("mimo::", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", k)
    return (m.group(1) + (m.group(2) or "")) if m else k[:60]


busy = 0.0
agg = {}
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    us = (e - s) / 1e3
    busy += us
    name = short(r["Kernel_Name"])
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += us
    if "--summary" not in sys.argv:
        grid = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        print(f"{i:4d} +{(s - t0) / 1e3:9.1f} us {us:8.1f} us  grid {grid:>9}  {name}")
span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
print(f"# {len(step)} launches, span {span / 1e3:.3f} ms, summed kernel time {busy / 1e3:.3f} ms, gaps {(span - busy) / 1e3:.3f} ms")
for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"# {us / 1e3:8.3f} ms  {n:4d} x {us / n:8.1f} us  {name}")
