"""Join a rocprofv3 --kernel-trace --pmc run of scripts/conv_layer_bench.py per conv dispatch:
duration, effective clock (GRBM_GUI_ACTIVE / 8 / duration), matrix-pipe busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles), instruction counts per MFMA.

    python3 scripts/layer_pmc.py <dir>
"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
cc = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"])
ctr = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    ctr[r["Dispatch_Id"]][r["Counter_Name"]] = ctr[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
labels = open(d + "/labels.txt").read().split("\n")[:-1]
ids = sorted((i for i in dur if ("conv3x3_" in dur[i][1] or "wgrad_split" in dur[i][1] or "wgrad_mfma" in dur[i][1])
              and "pack" not in dur[i][1]), key=int)
assert len(ids) == len(labels), (len(ids), len(labels))
seen = set()
for lab, i in zip(labels, ids):
    if lab in seen:
        continue  # first repetition only would be cold: keep the LAST one instead
    last = [j for l, j in zip(labels, ids) if l == lab][-1]
    seen.add(lab)
    us, k = dur[last]
    c = ctr[last]
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3 if us else 0  # GHz
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc if cyc else 0
    m = max(c.get("SQ_INSTS_MFMA", 0), 1)
    wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{lab:26s} {us:8.1f} us  clk {clk:4.2f} GHz  mfma_busy {busy:5.3f}  valu/mfma {c.get('SQ_INSTS_VALU', 0) / m:5.2f}  "
          f"wait_any {c.get('SQ_WAIT_ANY', 0) / wc:5.3f} wait_inst {c.get('SQ_WAIT_INST_ANY', 0) / wc:5.3f} "
          f"active {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.3f}  {k.split('(')[0].replace('void mimo::', '')}")
