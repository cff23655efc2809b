#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_order.txt; : > $O
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_ORDER_BY_PAIRS" "" "-DSWR_ORDER_BY_PAIRS" >> $O 2>&1
ABLATE_N=30 python3 tools/ablate.py cfg4 "" "-DSWR_ORDER_BY_PAIRS" >> $O 2>&1
ABLATE_N=30 python3 tools/ablate.py cfg2 "" "-DSWR_ORDER_BY_PAIRS" >> $O 2>&1
cat $O
