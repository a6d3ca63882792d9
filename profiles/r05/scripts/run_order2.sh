#!/bin/bash
# final tree: all GPU tests in two more shuffled orders
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_order2
mkdir -p $O
cd $R
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" > $O/ids.txt
for seed in ${SEEDS:-3 4}; do
  python3 - $O/ids.txt $seed > $O/ids_$seed.txt <<'P'
import random, sys
ids = [l.strip() for l in open(sys.argv[1]) if l.strip()]
random.Random(int(sys.argv[2])).shuffle(ids)
print("\n".join(ids))
P
  MIMO_PARITY_LOG=$O/parity_$seed.txt python -m pytest $(cat $O/ids_$seed.txt | tr '\n' ' ') -q -p no:cacheprovider 2>&1 | tail -4 > $O/shuffled_$seed.txt
done
tail -n 2 $O/shuffled_*.txt; grep -h host_fed_loop $O/parity_*.txt | cut -c55-460
