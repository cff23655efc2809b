#!/bin/bash
set -o pipefail
for p in 0 2; do for m in 0 2 3; do
  timeout -k 10 200 python tools/ab/evframes.py softwarerenderer_amd/libswr_hip.so cfg3 $p $m 2>&1 | tail -1 || exit 1
done; done
for st in 20; do for p in 0 2; do
  timeout -k 10 200 python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-profile-events --pipelining $p > gpurun_out/r4_fp.json 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_fp.json") if l.startswith("{")][-1])
print("bench cfg3 steps $st pipelining $p NO events: ms/step", j["ms_per_step"])
PY
done; done
