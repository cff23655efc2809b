#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for lib in build_ab/r4_blocked.so build_ab/r4_phongfast.so build_ab/r4_phongfast4.so build_ab/r4_blocked.so build_ab/r4_phongfast.so build_ab/r4_phongfast4.so; do
  timeout -k 10 200 python tools/ab/stages.py $lib cfg4 2>&1 | tail -1 | tee -a gpurun_out/r4_phong_ab.txt || exit 1
done
