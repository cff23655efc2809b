#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r5_segv.log
: > $L
run() { echo "=== $*" >> $L; timeout -k 10 600 python -X faulthandler -m pytest "$@" -x -q -m gpu -p no:cacheprovider >> $L 2>&1; echo "rc=$?" >> $L; }
run tests/test_gpu_cpp.py
run "tests/test_gpu_api.py::test_a_single_draw_beyond_the_batch_limit_is_refused_when_it_is_recorded" tests/test_gpu_cpp.py
run tests/test_gpu_api.py tests/test_gpu_cpp.py -k "not single_draw_beyond"
run tests/test_gpu_api.py tests/test_gpu_cpp.py
grep -E "^===|rc=|passed|failed|Segmentation|File \"/root/repo/tests" $L
which gdb catchsegv 2>&1 | head -2
