#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/profile_round.sh TAG
# One round's evidence, everything under gpurun_out/ (tools/summarize_rocprof.py then writes the tracked files under profiles/):
#   1. rocprofv3 --kernel-trace --stats of bench.py (cfg3 = the bench workload): per-kernel time
#   2. separate --pmc passes (the pool refuses --pmc together with other trace domains): FETCH_SIZE, WRITE_SIZE (HBM-side traffic),
#      and two SQ passes (instruction mix, VALU / LDS activity, waits, bank conflicts)
#   3. the plain bench line with the CPU baseline
#   4. kernel stats + bench lines of the other single-GPU configurations (cfg2, cfg4, cfg5)
TAG=${1:-r02}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
B="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline"
S="python3 bench.py --steps 3 --warmup 1 --prime 4 --no-cpu-baseline --no-profile-events"
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_sq1_$TAG gpurun_out/pmc_sq2_$TAG
python3 bench.py --print-src-hash > gpurun_out/srchash_$TAG.txt &&
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o run --output-format csv -- $B > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_$TAG.err &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$TAG -o run --output-format csv -- $S > /dev/null 2> gpurun_out/pmc_fetch_$TAG.err &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$TAG -o run --output-format csv -- $S > /dev/null 2> gpurun_out/pmc_write_$TAG.err &&
rocprofv3 --kernel-trace --pmc $P1 -d gpurun_out/pmc_sq1_$TAG -o run --output-format csv -- $S > /dev/null 2> gpurun_out/pmc_sq1_$TAG.err &&
rocprofv3 --kernel-trace --pmc $P2 -d gpurun_out/pmc_sq2_$TAG -o run --output-format csv -- $S > /dev/null 2> gpurun_out/pmc_sq2_$TAG.err &&
python3 tools/pmc_table.py gpurun_out/pmc_sq1_$TAG gpurun_out/pmc_sq2_$TAG > gpurun_out/sq_counters_$TAG.txt &&
echo "profiles of cfg3 done" &&
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err &&
tail -1 gpurun_out/bench_$TAG.json | cut -c1-300 &&
for c in cfg2 cfg4 cfg5; do
  rm -rf gpurun_out/prof_${TAG}_$c
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_$c -o run --output-format csv -- python3 bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${TAG}_$c.json 2> gpurun_out/prof_${TAG}_$c.err || exit 1
  echo "$c done"
done
