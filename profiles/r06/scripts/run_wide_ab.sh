#!/bin/bash
# round 6: per-layer convolution times with the wide kernel by rule (1), forced (2), never (0) — is the round-3 fit of sched::wide_config still right?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/ab_conv_layers.sh r06_wide_ab MIMO_CONV_WIDE 1 2
bash scripts/ab_conv_layers.sh r06_wide_ab MIMO_CONV_WIDE 0 0
ls gpurun_out/r06_wide_ab
