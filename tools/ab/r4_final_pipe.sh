#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for cfg in cfg3 cfg4 cfg5; do
for p in 0 2 0 2; do
  timeout -k 10 200 python tools/ab/frames.py softwarerenderer_amd/libswr_hip.so $cfg $p 2>&1 | tail -1 | tee -a gpurun_out/r4_final_pipe.txt || exit 1
done; done
for st in 20 50; do for p in 0 2 0 2; do
  timeout -k 10 200 python bench.py --steps $st --warmup 5 --no-cpu-baseline --pipelining $p > gpurun_out/r4_fp.json 2>/dev/null || exit 1
  python - <<PY | tee -a gpurun_out/r4_final_pipe.txt
import json
j=json.loads([l for l in open("gpurun_out/r4_fp.json") if l.startswith("{")][-1])
print("bench cfg3 steps $st pipelining $p: ms/step", j["ms_per_step"], "unpiped", j.get("ms_per_step_unpipelined"), "kernel", j["roofline"]["timed_region"]["kernel_ms_median"])
PY
done; done
