#!/bin/bash
# final evidence, part C: the sweep through every numerics-mode build against the oracle built alike
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
for v in fma dotpw fma_dotpw dpps; do
  SWR_LIB=libswr_hip_$v.so SWR_ORACLE_VARIANT=$v python3 tools/parity_sweep.py 110 21 > gpurun_out/sweep_r03_final_$v.log 2>&1 || { tail -5 gpurun_out/sweep_r03_final_$v.log; exit 1; }
  echo "$v: $(tail -1 gpurun_out/sweep_r03_final_$v.log)"
done
