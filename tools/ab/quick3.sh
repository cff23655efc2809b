#!/bin/bash
# usage: tools/ab/quick3.sh TAG "variant" ... -- parity + fullsize + test-lib + api tests on the in-tree build, then every stage of cfg3 (and cfg2 / cfg4 for the in-tree build)
cd "$GRAFT_REPO_ROOT"; TAG=${1:-q}; shift
export SWR_DEV_BUILD=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_testlib.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; RC=$?; tail -3 gpurun_out/gpu_tests_$TAG.log
export ABLATE_N=40
python tools/ablate.py cfg3 lib:build_ab/r2.so "" "$@" lib:build_ab/r2.so "" > gpurun_out/ab_$TAG.txt 2>&1
python tools/ablate.py cfg2 lib:build_ab/r2.so "" >> gpurun_out/ab_$TAG.txt 2>&1
python tools/ablate.py cfg4 lib:build_ab/r2.so "" >> gpurun_out/ab_$TAG.txt 2>&1
cat gpurun_out/ab_$TAG.txt
exit $RC
