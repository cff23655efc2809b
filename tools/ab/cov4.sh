#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_cov4.txt; : > $O
ABLATE_N=40 python3 tools/ablate.py cfg3 "" "-DSWR_COVER_EXCHANGE_R3" "" "-DSWR_COVER_EXCHANGE_R3" >> $O 2>&1
cat $O
