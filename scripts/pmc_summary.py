"""Summarise rocprofv3 --pmc counter CSVs per kernel (sum over dispatches)."""
import csv, sys, collections, glob, os
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void mimo::", "").replace("mimo::", "").split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], name)
        if key not in seen:
            seen.add(key); calls[name] += 1
names = sorted(agg, key=lambda n: -agg[n].get("SQ_WAVE_CYCLES", agg[n].get("FETCH_SIZE", 0)))
for n in names[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(n, "calls", calls[n], {k: ("%.3g" % v) for k, v in sorted(agg[n].items())})
