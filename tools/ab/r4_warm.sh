#!/bin/bash
set -o pipefail
for w in 5 40; do for p in 0 2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-profile-events --pipelining $p > gpurun_out/r4_fp.json 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_fp.json") if l.startswith("{")][-1])
print("bench cfg3 steps 20 warmup $w pipelining $p NO events: ms/step", j["ms_per_step"])
PY
done; done
for w in 5 40; do for p in 0 2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline --pipelining $p > gpurun_out/r4_fp.json 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_fp.json") if l.startswith("{")][-1])
print("bench cfg3 steps 20 warmup $w pipelining $p events: ms/step", j["ms_per_step"], "kernel", j["roofline"]["timed_region"]["kernel_ms_median"], "iso", j["roofline"].get("kernel_ms_isolated"))
PY
done; done
