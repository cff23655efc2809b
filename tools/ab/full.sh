#!/bin/bash
# usage: tools/ab/full.sh TAG -- whole GPU suite on the in-tree build (development: the build-identity test is skipped), then r2 vs in-tree on every config
cd "$GRAFT_REPO_ROOT"; TAG=${1:-full}
export SWR_DEV_BUILD=1
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1; RC=$?; tail -3 gpurun_out/gpu_tests_$TAG.log
export ABLATE_N=30
for c in cfg3 cfg2 cfg4 cfg5; do python tools/ablate.py $c lib:build_ab/r2.so lib:softwarerenderer_amd/libswr_hip.so lib:build_ab/r2.so lib:softwarerenderer_amd/libswr_hip.so > gpurun_out/ab_${TAG}_$c.txt 2>&1; done
python - <<P
import json,re
for c in ("cfg3","cfg2","cfg4","cfg5"):
    print(c)
    for ln in open('gpurun_out/ab_${TAG}_%s.txt' % c):
        m=re.match(r'(.*?)\s*(\{.*\})',ln)
        if m: d=json.loads(m.group(2)); print('  ', m.group(1)[:60].ljust(60), ' '.join(f"{k[:-3]}={v}" for k,v in d.items()))
        else: print(ln.rstrip()[:200])
P

python bench.py --config cfg3 --bilinear --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${TAG}_bilinear.json 2> gpurun_out/bench_${TAG}_bilinear.err; tail -c 1200 gpurun_out/bench_${TAG}_bilinear.json
exit $RC
