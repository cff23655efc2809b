#!/bin/bash
# whole GPU suite + the driver's bench line on the committed source
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_suite_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r5_suite_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_20.json 2> gpurun_out/r5_bench_20.err; rc=$?
python -c "import json; d=json.loads(open('gpurun_out/r5_bench_20.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['cpu_baseline']['value'])"
exit $rc
