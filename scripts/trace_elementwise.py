"""Per-launch durations of the bandwidth-class kernels from a rocprofv3 kernel trace (one training step).

    python scripts/trace_elementwise.py <dir with *_kernel_trace.csv> [name substrings ...]
Prints the launches of the last step in stream order: kernel, grid, duration.
"""
import csv
import glob
import sys

d = sys.argv[1]
names = sys.argv[2:] or ["upcat_fwd", "up_bwd", "pool_bwd", "fold_slice", "maxpool_fwd", "head_", "bn_", "bnrelu"]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = (adam[-2], adam[-1]) if len(adam) >= 2 else (0, len(rows))
agg = {}
for r in rows[lo:hi]:
    k = r["Kernel_Name"]
    if not any(n in k for n in names):
        continue
    short = k.split("(")[0].replace("mimo::", "").replace("void ", "")
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{short:28s} grid {r['Grid_Size_X']:>8s} x {r['Grid_Size_Y']:>4s}  {us:8.1f} us")
    agg[short] = agg.get(short, 0.0) + us
print({k: round(v, 1) for k, v in agg.items()})
