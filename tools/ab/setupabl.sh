#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_setupabl.txt; : > $O
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_SETUP_MINBLOCKS=7" "-DSWR_SETUP_MINBLOCKS=8" "" >> $O 2>&1
cat $O
