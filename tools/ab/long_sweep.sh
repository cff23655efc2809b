#!/bin/bash
# a long parity sweep on the in-tree build with OTHER seeds than the closing sweeps: default build, frames in flight, wireframe / banded
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
python3 tools/parity_sweep.py ${LONG_S:-420} ${SWEEP_SEED:-101} > gpurun_out/sweep_long.log 2>&1 || { tail -5 gpurun_out/sweep_long.log; exit 1; }
tail -1 gpurun_out/sweep_long.log
python3 tools/parity_sweep_pipeline.py ${LONG_P:-360} ${SWEEP_SEED:-101} > gpurun_out/sweep_long_pipeline.log 2>&1 || { tail -5 gpurun_out/sweep_long_pipeline.log; exit 1; }
tail -1 gpurun_out/sweep_long_pipeline.log
python3 tools/parity_sweep_modes.py ${LONG_M:-160} > gpurun_out/sweep_long_modes.log 2>&1 || { tail -5 gpurun_out/sweep_long_modes.log; exit 1; }
tail -2 gpurun_out/sweep_long_modes.log
