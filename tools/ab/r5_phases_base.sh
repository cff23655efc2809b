#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "== base (HEAD)" | tee -a gpurun_out/r5_phases.txt
timeout -k 10 400 python tools/phase_times.py cfg3 2>&1 | tail -10 | tee -a gpurun_out/r5_phases.txt
