#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for cfg in cfg3 cfg2; do
for lib in build_ab/r4_preretune.so build_ab/r4_retuned.so build_ab/r4_preretune.so build_ab/r4_retuned.so; do
  timeout -k 10 200 python tools/ab/stages.py $lib $cfg 2>&1 | tail -1 | tee -a gpurun_out/r4_stage_ab.txt || exit 1
done; done
