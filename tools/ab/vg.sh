#!/bin/bash
cd "$GRAFT_REPO_ROOT"; TAG=${1:-vg}
export SWR_DEV_BUILD=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; RC=$?; tail -3 gpurun_out/gpu_tests_$TAG.log
export ABLATE_N=30
python tools/ablate.py cfg3 lib:build_ab/r2.so "" "-DSWR_NO_FRAG_ALIAS" "-DSWR_ABL_LDSBYTES=2048" "-DSWR_VARY_GLOBAL=0" "-DSWR_VARY_GLOBAL=0 -DSWR_NO_FRAG_ALIAS" lib:build_ab/r2.so "" > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
for c in cfg2 cfg4; do python tools/ablate.py $c lib:build_ab/r2.so "" "-DSWR_VARY_GLOBAL=0" > gpurun_out/ab_${TAG}_$c.txt 2>&1; cat gpurun_out/ab_${TAG}_$c.txt; done
exit $RC
