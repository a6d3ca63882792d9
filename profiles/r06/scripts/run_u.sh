#!/bin/bash
# round 6, call U: overlapped timeline of one step at batch 32 (default configuration): where does the main stream wait?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/t -o b32 --output-format csv -- python3 $R/bench.py --steps 8 --warmup 4 --profile-steps 0 --no-cpu-baseline --no-strict > $O/bench_under_rocprof.json 2> $O/err.txt
cd $R
python3 scripts/trace_overlap.py $O/t > $O/overlap_b32.txt 2>&1
python3 - <<PY
import re
rows = []
for l in open("$O/overlap_b32.txt"):
    m = re.match(r"\s*(\d+) q(\S+) \+\s*([\d.]+)\s+([\d.]+) us  overlap\s+([\d.]+) us  (.*)", l)
    if m:
        rows.append((int(m.group(1)), m.group(2), float(m.group(3)), float(m.group(4)), float(m.group(5)), m.group(6)))
main = max(set(r[1] for r in rows), key=lambda q: sum(1 for r in rows if r[1] == q))
mq = [r for r in rows if r[1] == main]
gaps = []
for a, b in zip(mq, mq[1:]):
    g = b[2] - (a[2] + a[3])
    if g > 8.0:
        gaps.append((g, b[2], a[5][:50], b[5][:50]))
print("main queue", main, "kernels", len(mq), "busy %.3f ms" % (sum(r[3] for r in mq) / 1e3), "span %.3f ms" % ((mq[-1][2] + mq[-1][3] - mq[0][2]) / 1e3))
print("gaps > 8 us on the main queue: %d, total %.3f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e3))
for g in sorted(gaps, reverse=True)[:25]:
    print("  %7.1f us at +%9.1f  after %-50s before %s" % g)
side = [r for r in rows if r[1] != main]
print("side queue kernels", len(side), "busy %.3f ms" % (sum(r[3] for r in side) / 1e3))
PY
tail -4 $O/overlap_b32.txt
rm -rf $O/t
