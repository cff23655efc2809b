#!/bin/bash
cd "$GRAFT_REPO_ROOT"; TAG=${1:-flags}
export ABLATE_N=30
python tools/ablate.py cfg3 "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-early-ifcvt=1" "-mllvm -amdgpu-schedule-metric-bias=100" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-O2" "" > gpurun_out/ab_$TAG.txt 2>&1
python - <<P
import json,re
for ln in open('gpurun_out/ab_$TAG.txt'):
    m=re.match(r'(.*?)\s*(\{.*\})',ln)
    if m: d=json.loads(m.group(2)); print(m.group(1)[:70].ljust(70), 'cover', d['cover_ms'], 'raster', d['raster_ms'], 'total', d['total_ms'])
    else: print(ln.rstrip()[:200])
P
