#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "bilinear" 2>&1 | tail -3 || exit 1
for lib in build_ab/r4_binfree.so build_ab/r4_blocked.so build_ab/r4_binfree.so build_ab/r4_blocked.so; do
  timeout -k 10 200 python tools/ab/stages.py $lib cfg3_bilinear 2>&1 | tail -1 | tee -a gpurun_out/r4_bilinear_ab.txt || exit 1
done
for lib in build_ab/r4_blocked.so build_ab/r4_texfixed.so build_ab/r4_blocked.so build_ab/r4_texfixed.so; do
  timeout -k 10 200 python tools/ab/stages.py $lib cfg3 2>&1 | tail -1 | tee -a gpurun_out/r4_bilinear_ab.txt || exit 1
done
