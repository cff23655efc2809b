#!/bin/bash
# issue priority by phase inside k_raster_c: a = skeleton 1 / shading 0, b = skeleton 0 / shading 1, c = skeleton 2 / shading 0
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r5_prio_ab.txt
for cfg in cfg3 cfg2; do for rep in 1 2; do for lib in r5_head r5_prio_a r5_prio_b r5_prio_c; do
  timeout -k 10 200 python tools/ab/stages.py build_ab/$lib.so $cfg 2>&1 | tail -1 | tee -a $OUT || exit 1
done; done; done
