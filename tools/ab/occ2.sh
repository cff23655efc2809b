#!/bin/bash
cd "$GRAFT_REPO_ROOT"; TAG=${1:-occ2}
export ABLATE_N=30
python tools/ablate.py cfg3 "-DSWR_VARY_GLOBAL=0" "-DSWR_BATCH=28 -DSWR_WINDOW=56 -DSWR_RASTER_MINWAVES=4" "-DSWR_BATCH=24 -DSWR_WINDOW=48 -DSWR_RASTER_MINWAVES=4" "-DSWR_BATCH=20 -DSWR_WINDOW=40 -DSWR_RASTER_MINWAVES=4" "-DSWR_BATCH=28 -DSWR_WINDOW=64 -DSWR_RASTER_MINWAVES=4" "-DSWR_VARY_GLOBAL=0" > gpurun_out/ab_$TAG.txt 2>&1
python - <<'P'
import json,re
for ln in open('gpurun_out/ab_occ2.txt'):
    m=re.match(r'(.*?)\s*(\{.*\})',ln)
    if m: d=json.loads(m.group(2)); print(m.group(1)[:80].ljust(80), d['raster_ms'], d['total_ms'])
    else: print(ln.rstrip()[:200])
P
