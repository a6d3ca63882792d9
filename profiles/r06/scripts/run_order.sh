#!/bin/bash
# round 6: order dependence of the GPU suite on the final tree (every test file alone in a process, all tests in two shuffled orders),
# the stream stress tool and a 2000-step soak fed from host batches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_order
mkdir -p $O
cd $R
: > $O/per_file.txt
for f in tests/test_*gpu*.py; do
  echo "$f: $(python -m pytest $f -m gpu -q 2>&1 | tail -1)" >> $O/per_file.txt
done
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" > $O/ids.txt
for seed in 1 2; do
  python3 - $O/ids.txt $seed > $O/ids_$seed.txt <<'P'
import random, sys
ids = [l.strip() for l in open(sys.argv[1]) if l.strip()]
random.Random(int(sys.argv[2])).shuffle(ids)
print("\n".join(ids))
P
  python -m pytest $(cat $O/ids_$seed.txt | tr '\n' ' ') -q -p no:cacheprovider 2>&1 | tail -6 > $O/shuffled_$seed.txt
done
cat $O/per_file.txt; tail -2 $O/shuffled_1.txt $O/shuffled_2.txt
timeout 600 python tests/tools/soak.py 2000 > $O/soak.txt 2>&1; tail -8 $O/soak.txt
