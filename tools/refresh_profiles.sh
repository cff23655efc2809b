#!/bin/bash
# usage (here, after `gpurun -- bash tools/profile_round.sh TAG` has merged its output into gpurun_out/): bash tools/refresh_profiles.sh TAG OUT
# -> profiles/OUT_* (kernel stats + summaries of cfg3 / cfg2 / cfg4 / cfg5, SQ counter table, bench line) and profiles/<round>_traffic.json
TAG=${1:-r02}; OUT=${2:-r02_final}
cd "$(dirname "$0")/.." || exit 1
python3 tools/summarize_rocprof.py $OUT gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/prof_${TAG}_bench.json gpurun_out/srchash_$TAG.txt cfg3 gpurun_out/pmc_sq1_$TAG || exit 1
for c in cfg2 cfg4 cfg5; do
  python3 tools/summarize_rocprof.py ${OUT}_$c gpurun_out/prof_${TAG}_$c "" "" gpurun_out/bench_${TAG}_$c.json || exit 1
done
cp gpurun_out/sq_counters_$TAG.txt profiles/${OUT}_sq_counters.txt
# (the bench line of the profile round was printed before this script refreshed the traffic file: run `python bench.py` once more on
#  the GPU afterwards and store its line as profiles/OUT_bench_with_cpu.json -- it then carries roofline.traffic)
python3 bench.py --print-src-hash; grep -o '"csrc_sha256": "[0-9a-f]*"' profiles/*_traffic.json
