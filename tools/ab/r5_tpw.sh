#!/bin/bash
# round 5 (session 2 of round 4): tiles per raster wave (per-XCD tile queue).  libswr_hip_ab.so = this source with -DSWR_AB_ENV
# (SWR_AB_TPW overrides the host's choice), libswr_hip_base.so = the previous commit.
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r5_tpw_ab.txt
SWR_LIB=libswr_hip_ab.so SWR_AB_TPW=3 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r5_tpw_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5_tpw_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  timeout -k 10 200 python tools/ab/stages.py softwarerenderer_amd/libswr_hip_base.so cfg3 2>&1 | tail -1 | tee -a $OUT || exit 1
  for t in 1 2 4 8 16 64; do
    SWR_AB_TPW=$t timeout -k 10 200 python tools/ab/stages.py softwarerenderer_amd/libswr_hip_ab.so cfg3 2>&1 | tail -1 | sed "s/^/tpw=$t /" | tee -a $OUT || exit 1
  done
done
for t in 1 4 8; do
  SWR_AB_TPW=$t timeout -k 10 200 python tools/ab/stages.py softwarerenderer_amd/libswr_hip_ab.so cfg2 2>&1 | tail -1 | sed "s/^/tpw=$t /" | tee -a $OUT || exit 1
done
for t in 1 4 8; do
  SWR_LIB=libswr_hip_ab.so SWR_AB_TPW=$t timeout -k 10 300 python bench.py --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench tpw=$t', d['ms_per_step'], d.get('ms_per_step_unpipelined'), d['roofline']['kernel_ms'])" | tee -a $OUT || exit 1
done
