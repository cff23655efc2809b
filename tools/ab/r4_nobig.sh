#!/bin/bash
# timing-only experiment: k_bin without its big-triangle path (47 / 30 VGPRs: two / three waves per SIMD beside the raster kernel instead of
# one) -- what would a lean k_bin be worth with frames in flight?  (cfg3 / cfg5 have next to no triangle of more than 8 tiles.)
set -o pipefail
mkdir -p gpurun_out
for cfg in cfg3 cfg5; do for p in 1 0; do
for lib in softwarerenderer_amd/libswr_hip.so build_ab/r4_nobig.so softwarerenderer_amd/libswr_hip.so build_ab/r4_nobig.so; do
  timeout -k 10 200 python tools/ab/frames.py $lib $cfg $p 2>&1 | tail -1 | tee -a gpurun_out/r4_nobig.txt || exit 1
done; done; done
