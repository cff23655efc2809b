#!/bin/bash
# does an LDS-free front-end kernel run in the raster kernel's SPARE wave slots?  k_bin<COUNT> without its LDS table (global atomics
# per pair) against the product: alone (stages.py, one stream) and with frames in flight (bench.py, 50 steps)
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r5_nolds_ab.txt
for rep in 1 2; do for lib in head nolds; do
  timeout -k 10 200 python tools/ab/stages.py softwarerenderer_amd/libswr_hip_ab_$lib.so cfg3 2>&1 | tail -1 | tee -a $OUT || exit 1
done; done
for rep in 1 2; do for lib in head nolds; do
  SWR_LIB=libswr_hip_ab_$lib.so timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench $lib', d['ms_per_step'], d.get('ms_per_step_unpipelined'), d['roofline']['kernel_ms'], d['roofline']['timed_region']['kernel_ms_median'])" | tee -a $OUT || exit 1
done; done
for lib in head nolds; do
  SWR_LIB=libswr_hip_ab_$lib.so timeout -k 10 300 python bench.py --config cfg2 --steps 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench cfg2 $lib', d['ms_per_step'], d.get('ms_per_step_unpipelined'))" | tee -a $OUT || exit 1
done
