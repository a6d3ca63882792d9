#!/bin/bash
# host-fed loop at batch 4: why is the pinned feed 0.3 ms per step slower than the pageable one?  Kernel + memory-copy traces.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in pinned pageable; do
  rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/t_$m -o t --output-format csv -- python3 $R/bench.py --batch 4 --steps 40 --warmup 5 --profile-steps 0 --no-cpu-baseline --host-batches $m > $O/bench_$m.json 2> $O/err_$m.txt
  for f in $O/t_$m/*stats*.csv $O/t_$m/*domain*.csv; do [ -f $f ] && cp $f $O/${m}_$(basename $f); done
  python3 - $O/t_$m $m > $O/copies_$m.txt <<'P'
import csv, glob, sys
d, m = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/*memory_copy_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    print(m, len(rows), "copies; columns", list(rows[0].keys()) if rows else None)
    import collections
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r.get("Direction"), r.get("Bytes") or r.get("Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(agg.items(), key=lambda kv: -len(kv[1]))[:12]:
        v.sort()
        print(k, len(v), "median us", v[len(v) // 2] / 1e3, "max us", v[-1] / 1e3)
P
  rm -rf $O/t_$m
done
cd $R
for i in 1 2; do for m in pinned pageable; do for d in 2 3; do
  MIMO_PREFETCH_DEPTH=$d python3 bench.py --batch 4 --steps 60 --warmup 10 --profile-steps 0 --no-cpu-baseline --host-batches $m 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$m depth $d', l['value'], l['ms_per_step'])" >> $O/ab.txt
done; done
python3 bench.py --batch 4 --steps 60 --warmup 10 --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('resident', l['value'], l['ms_per_step'])" >> $O/ab.txt
done
cat $O/copies_*.txt $O/ab.txt
