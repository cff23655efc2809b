#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_coverabl.txt; : > $O
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_ABL_NOCOVERLOOP" "-DSWR_ABL_NOZBOUND" "-DSWR_ABL_NOCOVERLOOP -DSWR_ABL_NOZBOUND" "" >> $O 2>&1
cat $O
