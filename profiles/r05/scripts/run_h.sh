#!/bin/bash
# round-5 GPU call H: full GPU suite (range-safe weight images, arithmetic-specific bounds), serialised batch-4 timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_h
mkdir -p $O
cd "$R"
MIMO_PARITY_LOG=$O/parity_errors.txt timeout 2000 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
MIMO_WGRAD_STREAM=0 rocprofv3 --kernel-trace -d $O/b4trace -o t --output-format csv -- python3 $R/bench.py --batch 4 --steps 12 --warmup 5 --profile-steps 0 --no-cpu-baseline > $O/b4trace.json 2> $O/b4trace.err
python3 $R/scripts/trace_step.py $O/b4trace > $O/b4_step_serial.txt 2>&1
rm -rf $O/b4trace
grep "^#" $O/b4_step_serial.txt | head -40
