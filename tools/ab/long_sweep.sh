#!/bin/bash
# a long parity sweep on the in-tree build (other seeds than the closing sweeps): default build, then wireframe / banded
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
python3 tools/parity_sweep.py 600 ${SWEEP_SEED:-101} > gpurun_out/sweep_long.log 2>&1 || { tail -5 gpurun_out/sweep_long.log; exit 1; }
tail -1 gpurun_out/sweep_long.log
python3 tools/parity_sweep_modes.py 220 > gpurun_out/sweep_long_modes.log 2>&1 || { tail -5 gpurun_out/sweep_long_modes.log; exit 1; }
tail -2 gpurun_out/sweep_long_modes.log
