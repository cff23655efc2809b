#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_cb.txt; : > $O
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_COVER_BLOCK=128" "-DSWR_COVER_BLOCK=512" "" >> $O 2>&1
cat $O
