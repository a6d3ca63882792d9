"""Summarise rocprofv3 --pmc counter CSVs per kernel (sum over dispatches).

    python scripts/pmc_summary.py <rocprof output dir> [top_n] [--csv out.csv]

Prints one line per kernel; with --csv also writes `kernel,dispatches,<counter>...` rows (the
condensed form committed under profiles/)."""
import collections
import csv
import glob
import os
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
d = args[0]
top = int(args[1]) if len(args) > 1 else 14
out_csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], name)
        if key not in seen:
            seen.add(key)
            calls[name] += 1
counters = sorted({c for v in agg.values() for c in v})
names = sorted(agg, key=lambda n: -agg[n].get("SQ_WAVE_CYCLES", agg[n].get("FETCH_SIZE", agg[n].get(counters[0], 0))))
for n in names[:top]:
    short = n.replace("void mimo::", "").replace("mimo::", "").split("(")[0]
    print(short, "calls", calls[n], {k: ("%.4g" % v) for k, v in sorted(agg[n].items())})
if out_csv:
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches"] + counters)
        for n in names:
            w.writerow([n, calls[n]] + ["%.6g" % agg[n].get(c, 0.0) for c in counters])
