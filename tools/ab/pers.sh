#!/bin/bash
cd "$GRAFT_REPO_ROOT"; TAG=${1:-pers}
export SWR_DEV_BUILD=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; RC=$?; tail -3 gpurun_out/gpu_tests_$TAG.log
export ABLATE_N=30
for c in cfg3 cfg2 cfg4; do
python tools/ablate.py $c lib:build_ab/r2.so "env:SWR_RASTER_GRID=0" "env:SWR_RASTER_GRID=-1" "env:SWR_RASTER_GRID=20" "env:SWR_RASTER_GRID=32" "env:SWR_RASTER_GRID=0" "env:SWR_RASTER_GRID=-1" > gpurun_out/ab_${TAG}_$c.txt 2>&1
python - <<P
import json,re
print("$c")
for ln in open('gpurun_out/ab_${TAG}_$c.txt'):
    m=re.match(r'(.*?)\s*(\{.*\})',ln)
    if m: d=json.loads(m.group(2)); print('  ', m.group(1)[:40].ljust(40), d['raster_ms'], d['total_ms'])
    else: print(ln.rstrip()[:200])
P
done
exit $RC
