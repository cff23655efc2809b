#!/bin/bash
# closing verification + profiles of the round's second session (chain replay with the predicate in EXEC), one call, most important first
TAG=r04g
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
SWR_DEV_BUILD=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/sweep_${TAG}_tests.log 2>&1 || { tail -8 gpurun_out/sweep_${TAG}_tests.log; exit 1; }
tail -2 gpurun_out/sweep_${TAG}_tests.log
python3 tools/parity_sweep.py 130 53 > gpurun_out/sweep_${TAG}.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}.log; exit 1; }
tail -1 gpurun_out/sweep_${TAG}.log
python3 tools/parity_sweep_pipeline.py 100 53 > gpurun_out/sweep_${TAG}_pipeline.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}_pipeline.log; exit 1; }
tail -1 gpurun_out/sweep_${TAG}_pipeline.log
python3 tools/parity_sweep_modes.py 70 > gpurun_out/sweep_${TAG}_modes.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}_modes.log; exit 1; }
tail -2 gpurun_out/sweep_${TAG}_modes.log
SWR_LIB=libswr_hip_test.so python3 tools/parity_sweep.py 50 54 > gpurun_out/sweep_${TAG}_testlib.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}_testlib.log; exit 1; }
tail -1 gpurun_out/sweep_${TAG}_testlib.log
for v in fma dotpw fma_dotpw dpps fma_dpps; do
  SWR_LIB=libswr_hip_$v.so SWR_ORACLE_VARIANT=$v python3 tools/parity_sweep.py 20 55 > gpurun_out/sweep_${TAG}_$v.log 2>&1 || { tail -5 gpurun_out/sweep_${TAG}_$v.log; exit 1; }
  echo "$v: $(tail -1 gpurun_out/sweep_${TAG}_$v.log)"
done
echo "verification done"
bash tools/profile_round.sh r04g 2>&1 | tail -6
