#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_sortabl.txt; : > $O
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_ABL_SORT_NONET" "-DSWR_ABL_SORT_NOBIG" "-DSWR_ABL_SORT_NONET -DSWR_ABL_SORT_NOBIG" "-DSWR_ABL_SORT_NONET -DSWR_ABL_SORT_NOBIG -DSWR_ABL_SORT_NOPT" "" >> $O 2>&1
cat $O
