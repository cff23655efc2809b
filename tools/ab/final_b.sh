#!/bin/bash
# final evidence, part B (after tools/refresh_profiles.sh wrote the traffic file of this source): the bench line with live traffic, the
# bilinear line, and the parity sweeps on the final build (default build, wireframe / banded, the test library)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || exit 1
tail -1 gpurun_out/bench_final.json | cut -c1-200
python3 bench.py --config cfg3 --bilinear --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_final_bilinear.json 2> gpurun_out/bench_final_bilinear.err || exit 1
python3 tools/parity_sweep.py 330 7 > gpurun_out/sweep_r03_final.log 2>&1 || { tail -5 gpurun_out/sweep_r03_final.log; exit 1; }
tail -2 gpurun_out/sweep_r03_final.log
python3 tools/parity_sweep_modes.py 150 > gpurun_out/sweep_r03_final_modes.log 2>&1 || { tail -5 gpurun_out/sweep_r03_final_modes.log; exit 1; }
tail -3 gpurun_out/sweep_r03_final_modes.log
SWR_LIB=libswr_hip_test.so python3 tools/parity_sweep.py 150 13 > gpurun_out/sweep_r03_final_testlib.log 2>&1 || { tail -5 gpurun_out/sweep_r03_final_testlib.log; exit 1; }
tail -2 gpurun_out/sweep_r03_final_testlib.log
