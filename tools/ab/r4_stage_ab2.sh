#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for cfg in cfg3 cfg2; do
for lib in build_ab/r4_preretune.so build_ab/r4_binfree.so build_ab/r4_preretune.so build_ab/r4_binfree.so; do
  timeout -k 10 200 python tools/ab/stages.py $lib $cfg 2>&1 | tail -1 | tee -a gpurun_out/r4_stage_ab2.txt || exit 1
done; done
for cfg in cfg3 cfg2 cfg4; do
for lib in build_ab/r4_preretune.so build_ab/r4_binfree.so build_ab/r4_preretune.so build_ab/r4_binfree.so; do
for p in 0 1; do
  timeout -k 10 200 python tools/ab/frames.py $lib $cfg $p 2>&1 | tail -1 | tee -a gpurun_out/r4_stage_ab2.txt || exit 1
done; done; done
