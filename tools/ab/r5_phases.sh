#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r5_phases.txt
for t in 1 2 4; do
  echo "== static tiles per wave = $t" | tee -a $OUT
  SWR_AB_TPW=$t timeout -k 10 400 python tools/phase_times.py cfg3 -DSWR_AB_ENV 2>&1 | tail -10 | tee -a $OUT || exit 1
done
