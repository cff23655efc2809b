#!/bin/bash
# round 4, first GPU call: the suite on the pipelined library, then the bench with frames in flight on / off
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_api.py::test_build_identity_matches_the_verified_pair > gpurun_out/r4_tests1.log 2>&1; rc=$?
tail -5 gpurun_out/r4_tests1.log
[ $rc -ne 0 ] && exit $rc
for p in 1 0 2 1 0; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelining $p > gpurun_out/r4_b_cfg3_p$p.json 2>gpurun_out/r4_b_cfg3_p$p.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_b_cfg3_p$p.json") if l.startswith("{")][-1])
r=j["roofline"]
print("cfg3 p=$p ms/step", j["ms_per_step"], "unpiped", j.get("ms_per_step_unpipelined"), "kernel med/min/max", r["timed_region"]["kernel_ms_median"], r["timed_region"]["kernel_ms_min"], r["timed_region"]["kernel_ms_max"], "iso", r.get("kernel_ms_isolated"), "stages", r["stage_ms_per_step"])
PY
done
for p in 1 0; do
  timeout -k 10 200 python bench.py --config cfg2 --steps 50 --warmup 5 --no-cpu-baseline --pipelining $p > gpurun_out/r4_b_cfg2_p$p.json 2>gpurun_out/r4_b_cfg2_p$p.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_b_cfg2_p$p.json") if l.startswith("{")][-1])
r=j["roofline"]
print("cfg2 p=$p ms/step", j["ms_per_step"], "unpiped", j.get("ms_per_step_unpipelined"), "kernel med", r["timed_region"]["kernel_ms_median"], "iso", r.get("kernel_ms_isolated"))
PY
done
