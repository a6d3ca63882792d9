"""Per kernel name of a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES run of bench.py: launches, summed
duration, effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and matrix-pipe busy fraction.

    python3 scripts/step_clock.py <dir> [min_total_us]
"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
cc = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0].replace("void mimo::", ""))
ctr = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    ctr[r["Dispatch_Id"]][r["Counter_Name"]] = ctr[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for i, (us, k) in dur.items():
    a = agg[k]
    a[0] += 1
    a[1] += us
    a[2] += ctr[i].get("GRBM_GUI_ACTIVE", 0.0) / 8
    a[3] += ctr[i].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024
for k, (n, us, cyc, busy) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us < min_us:
        continue
    print(f"{n:5d} x {us / n:8.1f} us  total {us / 1e3:8.3f} ms  clk {cyc / us / 1e3 if us else 0:4.2f} GHz  mfma_busy {busy / cyc if cyc else 0:5.3f}  {k[:80]}")
