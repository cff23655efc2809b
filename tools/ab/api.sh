#!/bin/bash
cd "$GRAFT_REPO_ROOT"; TAG=${1:-api}
export SWR_DEV_BUILD=1
python -m pytest tests/test_gpu_api.py tests/test_gpu_multiproc.py -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; RC=$?; tail -15 gpurun_out/gpu_tests_$TAG.log
python tools/pcie_rate.py cfg3 > gpurun_out/pcie_$TAG.json 2> gpurun_out/pcie_$TAG.err; cat gpurun_out/pcie_$TAG.json; tail -3 gpurun_out/pcie_$TAG.err
exit $RC
