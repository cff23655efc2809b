#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_cfg2x.txt; : > $O
ABLATE_N=40 python3 tools/ablate.py cfg2 lib:build_ab/r2.so "" "-DSWR_SORT_INDEX_ORDER" lib:build_ab/r2.so "" "-DSWR_SORT_INDEX_ORDER" "" >> $O 2>&1
cat $O
