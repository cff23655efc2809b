#!/bin/bash
# upper bounds for the lookup / election part of k_raster_c (wrong images by design)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/ab_lookupabl.txt; : > $O
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_ABL_NOSELECT2" "-DSWR_ABL_NOSELECT" "-DSWR_ABL_NOELECT" "-DSWR_ABL_NOSHADE" "" >> $O 2>&1
cat $O
