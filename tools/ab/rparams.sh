#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_rparams.txt; : > $O
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_WINDOW=40" "-DSWR_WINDOW=48" "-DSWR_WINDOW=64" "-DSWR_WINDOW=24" "-DSWR_BATCH_FRAGS=1536" "" >> $O 2>&1
cat $O
