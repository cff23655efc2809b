#!/bin/bash
cd "$GRAFT_REPO_ROOT"; TAG=${1:-t}
export SWR_DEV_BUILD=1
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; RC=$?; tail -12 gpurun_out/gpu_tests_$TAG.log
exit $RC
