"""Per-kernel time of the last step in two rocprofv3 kernel traces, side by side.

    python scripts/trace_diff.py <dir A> <dir B>"""
import csv
import glob
import sys


def load(d):
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    agg = {}
    for r in rows[adam[-2]:adam[-1]]:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mimo::", "")[:60]
        agg[k] = agg.get(k, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg


a, b = load(sys.argv[1]), load(sys.argv[2])
tot_a = tot_b = 0.0
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, 0) + b.get(k, 0))):
    x, y = a.get(k, 0.0), b.get(k, 0.0)
    tot_a += x
    tot_b += y
    if abs(x - y) > 5.0:
        print(f"{k:62s} {x:9.1f} {y:9.1f} {y - x:+8.1f}")
print(f"{'total':62s} {tot_a:9.1f} {tot_b:9.1f} {tot_b - tot_a:+8.1f}")
