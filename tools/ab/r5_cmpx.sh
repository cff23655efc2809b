#!/bin/bash
# chain replay with the predicate in EXEC (v_cmpx per step) instead of three v_cndmask per step: parity first, then stages A/B
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r5_cmpx_ab.txt
SWR_LIB=libswr_hip_ab_cmpx.so timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r5_cmpx_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_cmpx_tests.log
[ $rc -eq 0 ] || exit $rc
for cfg in cfg3 cfg2 cfg4 cfg5; do for rep in 1 2; do for lib in r5_head r5_cmpx; do
  timeout -k 10 200 python tools/ab/stages.py build_ab/$lib.so $cfg 2>&1 | tail -1 | tee -a $OUT || exit 1
done; done; done
