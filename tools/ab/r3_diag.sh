#!/bin/bash
# round 3 diagnostic call: counters available, GPU suite on the round-2 kernels, occupancy / texel-latency probes of k_raster_c
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
(rocprofv3 -L > gpurun_out/r3_counters_avail.txt 2>&1 || rocprofv3 --list-avail > gpurun_out/r3_counters_avail.txt 2>&1 || true)
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests_base.log 2>&1; tail -2 gpurun_out/r3_gpu_tests_base.log
export ABLATE_N=30
python tools/ablate.py cfg3 "" "-DSWR_ABL_LDSBYTES=3456" "-DSWR_ABL_LDSBYTES=8192" "-DSWR_ABL_TEXFIXED" "-DSWR_ABL_NOTEX" "-DSWR_ABL_NOSHADE" "-DSWR_ABL_NOSHADE -DSWR_ABL_LDSBYTES=3456" "-DSWR_ABL_NOSHADE -DSWR_ABL_LDSBYTES=8192" "" > gpurun_out/r3_diag_ab.txt 2>&1; cat gpurun_out/r3_diag_ab.txt
