#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_api.py::test_build_identity_matches_the_verified_pair > gpurun_out/r4_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r4_tests.log
exit $rc
