#!/bin/bash
# the closing verification of a round's final source: whole GPU suite, then the sweeps (default build, wireframe / banded, frames in
# flight, the fenced test build, the numerics builds against the oracle built alike).  TAG names the logs under gpurun_out/.
TAG=${1:-r04}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
SWR_DEV_BUILD=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/sweep_${TAG}_tests.log 2>&1 || { tail -8 gpurun_out/sweep_${TAG}_tests.log; exit 1; }
tail -2 gpurun_out/sweep_${TAG}_tests.log
python3 tools/parity_sweep.py ${SWEEP_S:-240} ${SWEEP_SEED:-41} > gpurun_out/sweep_${TAG}.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}.log; exit 1; }
tail -1 gpurun_out/sweep_${TAG}.log
python3 tools/parity_sweep_modes.py ${SWEEP_M:-120} > gpurun_out/sweep_${TAG}_modes.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}_modes.log; exit 1; }
tail -2 gpurun_out/sweep_${TAG}_modes.log
python3 tools/parity_sweep_pipeline.py ${SWEEP_P:-200} ${SWEEP_SEED:-41} > gpurun_out/sweep_${TAG}_pipeline.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}_pipeline.log; exit 1; }
tail -1 gpurun_out/sweep_${TAG}_pipeline.log
SWR_LIB=libswr_hip_test.so python3 tools/parity_sweep.py ${SWEEP_T:-90} 43 > gpurun_out/sweep_${TAG}_testlib.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}_testlib.log; exit 1; }
tail -1 gpurun_out/sweep_${TAG}_testlib.log
for v in fma dotpw fma_dotpw dpps fma_dpps; do
  SWR_LIB=libswr_hip_$v.so SWR_ORACLE_VARIANT=$v python3 tools/parity_sweep.py ${SWEEP_V:-45} 44 > gpurun_out/sweep_${TAG}_$v.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}_$v.log; exit 1; }
  echo "$v: $(tail -1 gpurun_out/sweep_${TAG}_$v.log)"
done
