#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_driver.json 2> gpurun_out/r4_bench_driver.err || { tail -20 gpurun_out/r4_bench_driver.err; exit 1; }
tail -c 6000 gpurun_out/r4_bench_driver.json
for cfg in cfg3 cfg2; do for j in 0 0.05; do
  timeout -k 10 200 python bench.py --config $cfg --steps 50 --warmup 5 --no-cpu-baseline --camera-jitter $j > gpurun_out/r4_jit_${cfg}_$j.json 2>/dev/null || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_jit_${cfg}_$j.json") if l.startswith("{")][-1])
r=j["roofline"]
print("$cfg jitter $j: ms/step", j["ms_per_step"], "unpiped", j.get("ms_per_step_unpipelined"), "value", j["value"], "frags", j["fragments_tested_per_frame"], "kernel med", r["timed_region"]["kernel_ms_median"], "iso", r.get("kernel_ms_isolated"), "frac", r["frac"], r["timed_region"]["frac"])
PY
done; done
