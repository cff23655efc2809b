#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for cfg in cfg3 cfg2 cfg5; do
for lib in build_ab/r4_window.so build_ab/r4_coverscan.so build_ab/r4_window.so build_ab/r4_coverscan.so; do
  timeout -k 10 200 python tools/ab/stages.py $lib $cfg 2>&1 | tail -1 | tee -a gpurun_out/r4_cover_ab.txt || exit 1
done; done
