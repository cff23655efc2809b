#!/bin/bash
# static consecutive tiles per wave; _vc = per-lane VGPR counters, _lc = counters in LDS
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r5_tpw2_ab.txt
SWR_LIB=libswr_hip_ab_lc.so SWR_AB_TPW=3 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r5_tpw2_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_tpw2_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  timeout -k 10 200 python tools/ab/stages.py softwarerenderer_amd/libswr_hip_base.so cfg3 2>&1 | tail -1 | tee -a $OUT || exit 1
  for v in vc lc; do for t in 1 2 4 8; do
    SWR_AB_TPW=$t timeout -k 10 200 python tools/ab/stages.py softwarerenderer_amd/libswr_hip_ab_$v.so cfg3 2>&1 | tail -1 | sed "s/^/tpw=$t /" | tee -a $OUT || exit 1
  done; done
done
