#!/bin/bash
# the bench lines a round keeps under profiles/ (run on the GPU box AFTER tools/refresh_profiles.sh wrote the traffic file of this build):
# default (with the CPU baseline), the driver's form (20 steps), bilinear, and cfg2 / cfg3 under a moving camera.  TAG names the outputs.
TAG=${1:-r04}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py > gpurun_out/bench_${TAG}_final.json 2> gpurun_out/bench_${TAG}_final.err || exit 1
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${TAG}_driver.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --bilinear --no-cpu-baseline > gpurun_out/bench_${TAG}_bilinear.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --config cfg2 --camera-jitter 0.05 --no-cpu-baseline > gpurun_out/bench_${TAG}_cfg2_jitter.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --camera-jitter 0.05 --no-cpu-baseline > gpurun_out/bench_${TAG}_cfg3_jitter.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --pipelining 0 --no-cpu-baseline > gpurun_out/bench_${TAG}_driver_one_stream.json 2>/dev/null || exit 1
for f in final driver bilinear cfg2_jitter cfg3_jitter driver_one_stream; do
  python3 - <<PY
import json
j=json.loads([l for l in open("gpurun_out/bench_${TAG}_$f.json") if l.startswith("{")][-1])
r=j["roofline"]
print("$f: ms/step", j["ms_per_step"], "one stream", j.get("ms_per_step_unpipelined"), "value", j["value"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "timed region", r["timed_region"]["kernel_ms_median"], "traffic", r["traffic"])
PY
done
