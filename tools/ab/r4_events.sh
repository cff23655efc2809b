#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for lib in build_ab/r4_ev_default.so build_ab/r4_ev_nofence.so build_ab/r4_ev_dev.so build_ab/r4_ev_default.so build_ab/r4_ev_nofence.so build_ab/r4_ev_dev.so; do
for p in 1 0; do for m in 2 0; do
  timeout -k 10 200 python tools/ab/evframes.py $lib cfg3 $p $m 2>&1 | tail -1 | tee -a gpurun_out/r4_events_ab.txt || exit 1
done; done; done
