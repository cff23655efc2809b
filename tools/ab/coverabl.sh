#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_coverabl.txt; : > $O
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_ABL_NOCOVERLOOP" "-DSWR_ABL_NOZBOUND" "-DSWR_ABL_NOCOVERLOOP -DSWR_ABL_NOZBOUND" "-DSWR_COVER_HQ=1 -DSWR_COVER_WQ=4" "-DSWR_COVER_HQ=2 -DSWR_COVER_WQ=4" "" >> $O 2>&1
cat $O
