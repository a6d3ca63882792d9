#!/bin/bash
# round 6, call Y2: kernel times of the one-launch BatchNorm backward vs the three launches (serial streams, 4 images)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_y
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  MIMO_BN_BWD_SMALL=$v MIMO_WGRAD_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr$v -o t --output-format csv -- python3 $R/bench.py --batch 4 --steps 20 --warmup 5 --no-strict --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
  f=$(find $O/tr$v -name "*kernel_stats.csv" | head -1)
  echo "== MIMO_BN_BWD_SMALL=$v  $f" | tee -a $O/kernels.txt
  [ -n "$f" ] && grep -i "bn_bwd\|bnrelu_bwd" $f | cut -c1-260 | tee -a $O/kernels.txt
  [ -n "$f" ] && head -1 $f | tee -a $O/kernels.txt
  t=$(find $O/tr$v -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 - "$t" <<'PY' | tee -a $O/kernels.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full step: print the sequence of kernels between the last two adam kernels, with durations and gaps
idx = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
if len(idx) >= 2:
    a, b = idx[-2], idx[-1]
    prev_end = None
    for r in rows[a:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        name = r["Kernel_Name"].split("(")[0][-60:]
        print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8}  {name}")
        prev_end = e
PY
  find $O/tr$v -name "*kernel_trace.csv" -delete
  find $O/tr$v -name "*.db" -delete
done
