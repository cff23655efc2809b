#!/bin/bash
# usage: tools/ab/quick.sh TAG [cfg] -- development A/B on one box: parity + full-size + api tests on the in-tree build, then per-stage
# times of cfg (default cfg3) for build_ab/r2.so (round 2's final kernels) and the in-tree library, interleaved twice
cd "$GRAFT_REPO_ROOT"; TAG=${1:-x}; CFG=${2:-cfg3}
export SWR_DEV_BUILD=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; RC=$?; tail -3 gpurun_out/gpu_tests_$TAG.log
export ABLATE_N=30
python tools/ablate.py $CFG lib:build_ab/r2.so lib:softwarerenderer_amd/libswr_hip.so lib:build_ab/r2.so lib:softwarerenderer_amd/libswr_hip.so > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
exit $RC
