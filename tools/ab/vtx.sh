#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export SWR_DEV_BUILD=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/gpu_tests_vtx.log 2>&1; tail -2 gpurun_out/gpu_tests_vtx.log
O=gpurun_out/ab_vtx.txt; : > $O
ABLATE_N=40 python3 tools/ablate.py cfg3 "" "-DSWR_VERTEX_DIRECT_LOADS" "" "-DSWR_VERTEX_DIRECT_LOADS" >> $O 2>&1
cat $O
