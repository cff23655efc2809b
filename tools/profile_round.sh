#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/profile_round.sh TAG
# One round's evidence: rocprofv3 kernel stats of bench.py, separate PMC passes for HBM traffic (FETCH_SIZE, WRITE_SIZE:
# --pmc only together with --kernel-trace, as the pool requires), and the plain bench line with the CPU baseline.
# Everything lands under gpurun_out/; tools/summarize_rocprof.py then writes the tracked files under profiles/.
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o run --output-format csv -- $B > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_$TAG.err &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$TAG -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --prime 4 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_fetch_$TAG.err &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$TAG -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --prime 4 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_write_$TAG.err &&
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err &&
tail -1 gpurun_out/bench_$TAG.json
