"""How much kernels of different streams overlap in a rocprofv3 kernel trace: per kernel of the last full step its
stream / queue, start, end, and the time it shares with kernels of OTHER queues.

    python scripts/trace_overlap.py <dir with *_kernel_trace.csv> [first_row last_row]"""
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
qkey = "Queue_Id" if "Queue_Id" in step[0] else ("Stream_Id" if "Stream_Id" in step[0] else None)


def short(k):
    k = k.replace("void ", "").replace("mimo::", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", k)
    return (m.group(1) + (m.group(2) or "")) if m else k[:60]


iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(qkey, "?") if qkey else "?", short(r["Kernel_Name"])) for r in step]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, len(iv))
tot_ov = 0.0
for i, (s, e, q, name) in enumerate(iv):
    ov = 0
    for j, (s2, e2, q2, _) in enumerate(iv):
        if j != i and q2 != q:
            ov += max(0, min(e, e2) - max(s, s2))
    tot_ov += ov / 2e3
    if lo <= i < hi:
        print(f"{i:4d} q{q} +{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  overlap {ov / 1e3:7.1f} us  {name}")
span = (iv[-1][1] - t0) / 1e3
busy = sum(e - s for s, e, _, _ in iv) / 1e3
print(f"# columns of the trace: {list(step[0].keys())}")
print(f"# span {span / 1e3:.3f} ms, summed kernel time {busy / 1e3:.3f} ms, pairwise cross-queue overlap {tot_ov / 1e3:.3f} ms")

# GPU idle inside the step: gaps of the union of all kernel intervals (both streams), the largest ones with what follows
ivs = sorted((s, e, name) for s, e, _, name in iv)
idle, cur_end, gaps = 0, ivs[0][1], []
for s, e, name in ivs[1:]:
    if s > cur_end:
        idle += s - cur_end
        gaps.append((s - cur_end, (cur_end - t0) / 1e3, name))
    cur_end = max(cur_end, e)
print(f"# GPU idle inside the step (no kernel of either stream running): {idle / 1e6:.3f} ms in {len(gaps)} gaps")
for g, at, name in sorted(gaps, reverse=True)[:8]:
    print(f"#   {g / 1e3:7.1f} us idle at +{at:9.1f} us, before {name}")
