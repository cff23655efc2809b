#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for p in 1 0; do
for st in 20 50; do
  timeout -k 10 200 python bench.py --steps $st --warmup 5 --no-cpu-baseline --pipelining $p > gpurun_out/r4_bc.json 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_bc.json") if l.startswith("{")][-1])
print("bench steps $st p=$p events: ms/step", j["ms_per_step"], "unpiped", j.get("ms_per_step_unpipelined"), "kernel", j["roofline"]["timed_region"]["kernel_ms_median"])
PY
  timeout -k 10 200 python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-profile-events --pipelining $p > gpurun_out/r4_bc.json 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_bc.json") if l.startswith("{")][-1])
print("bench steps $st p=$p no events: ms/step", j["ms_per_step"])
PY
done; done; done
for p in 1 0; do for m in 2 0; do
  timeout -k 10 200 python tools/ab/evframes.py softwarerenderer_amd/libswr_hip.so cfg3 $p $m 2>&1 | tail -1 || exit 1
done; done
