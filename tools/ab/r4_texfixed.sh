#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for lib in build_ab/r4_blocked.so build_ab/r4_texfixed.so build_ab/r4_noshade.so build_ab/r4_blocked.so build_ab/r4_texfixed.so build_ab/r4_noshade.so; do
  timeout -k 10 200 python tools/ab/stages.py $lib cfg3 2>&1 | tail -1 | tee -a gpurun_out/r4_texfixed_ab.txt || exit 1
done
