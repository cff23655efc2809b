#!/bin/bash
# usage: tools/pmc_run.sh TAG [cfg] -- SQ counter passes over bench.py's frames (separate --pmc runs, kernel-trace only), summarised by pmc_table.py
TAG=${1:-run}; CFG=${2:-cfg3}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
rm -rf gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2
rocprofv3 --kernel-trace --pmc $P1 -d gpurun_out/pmc_${TAG}_1 -o p1 --output-format csv -- python3 bench.py --config $CFG --steps 10 --warmup 2 --prime 0 --no-cpu-baseline --no-profile-events > gpurun_out/pmc_${TAG}_1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc $P2 -d gpurun_out/pmc_${TAG}_2 -o p2 --output-format csv -- python3 bench.py --config $CFG --steps 10 --warmup 2 --prime 0 --no-cpu-baseline --no-profile-events > gpurun_out/pmc_${TAG}_2.log 2>&1 &&
python3 tools/pmc_table.py gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 --kernel k_raster > gpurun_out/pmc_${TAG}.txt && cat gpurun_out/pmc_${TAG}.txt
