#!/bin/bash
# bound of an LDS bank swizzle for the tile's colour / Z: -DSWR_ABL_NOCONFLICT (wrong image) against the product source
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r5_noconf_ab.txt
for cfg in cfg3 cfg2; do for rep in 1 2; do for lib in r5_head r5_noconf; do
  timeout -k 10 200 python tools/ab/stages.py build_ab/$lib.so $cfg 2>&1 | tail -1 | tee -a $OUT || exit 1
done; done; done
