#!/bin/bash
# usage: tools/ab/quick2.sh TAG "variant" "variant" ... -- parity+fullsize+api tests on the in-tree build, then ablate cfg3 over r2.so, the in-tree build and the variants
cd "$GRAFT_REPO_ROOT"; TAG=${1:-q}; shift
export SWR_DEV_BUILD=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_testlib.py -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; RC=$?; tail -3 gpurun_out/gpu_tests_$TAG.log
export ABLATE_N=40
python tools/ablate.py cfg3 lib:build_ab/r2.so "" "$@" lib:build_ab/r2.so "" > gpurun_out/ab_$TAG.txt 2>&1
python - <<P
import json,re
for ln in open('gpurun_out/ab_$TAG.txt'):
    m=re.match(r'(.*?)\s*(\{.*\})',ln)
    if m:
        try: d=json.loads(m.group(2)); print(m.group(1)[:70].ljust(70), 'cover', d['cover_ms'], 'raster', d['raster_ms'], 'total', d['total_ms'])
        except Exception: print(ln.rstrip()[:200])
    else: print(ln.rstrip()[:200])
P
exit $RC
