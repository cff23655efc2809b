#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_binabl.txt; : > $O
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_ABL_NOPMC" "-DSWR_BIN_TABLE_LOG2=10" "-DSWR_BIN_TABLE_LOG2=8" "" >> $O 2>&1
cat $O
