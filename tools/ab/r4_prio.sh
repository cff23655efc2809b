#!/bin/bash
# A/B: s_setprio N at the head of every front-end kernel (build_ab/r4_prioN.so, -DSWR_FRONT_PRIO=N of a scratch copy) -- with frames in
# flight the front end of frame N+1 is the critical path (547 us stretched across a 492 us raster kernel): does issue priority shorten it?
set -o pipefail
mkdir -p gpurun_out
for cfg in cfg3 cfg2 cfg5; do
for lib in softwarerenderer_amd/libswr_hip.so build_ab/r4_prio1.so build_ab/r4_prio3.so softwarerenderer_amd/libswr_hip.so build_ab/r4_prio1.so build_ab/r4_prio3.so; do
  timeout -k 10 200 python tools/ab/frames.py $lib $cfg 1 2>&1 | tail -1 | tee -a gpurun_out/r4_prio.txt || exit 1
done; done
