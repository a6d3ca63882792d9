#!/bin/bash
# the release-event race of the max |dz| slots: the new regression test and the test that exposed it, each ALONE in a process,
# three times, on the build with the event in front of the reduction (old) and behind it (fixed); then the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_z
mkdir -p $O
cd $R
for i in 1 2 3; do
  for v in old fixed; do
    if [ $v = old ]; then export MIMO_HIP_LIB=$R/build/variants/libmimo_oldevent.so; else unset MIMO_HIP_LIB; fi
    a=$(python -m pytest tests/test_network_gpu.py -m gpu -q -k "small_odd_geometry" 2>&1 | tail -1)
    b=$(python -m pytest tests/test_network_gpu.py -m gpu -q -k "pool_and_head" 2>&1 | tail -1)
    echo "$v run $i: regression test: $a | pool_and_head alone: $b" >> $O/race.txt
  done
done
unset MIMO_HIP_LIB
python -m pytest tests/test_data_gpu.py -m gpu -q 2>&1 | tail -3 > $O/pytest.txt
cat $O/race.txt $O/pytest.txt
