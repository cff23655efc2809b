#!/bin/bash
cd "$GRAFT_REPO_ROOT"; TAG=${1:-occ3}
export ABLATE_N=30
python tools/ablate.py cfg3 "-DSWR_VARY_GLOBAL=0" "-DSWR_VARY_GLOBAL=0 -DSWR_ABL_VARY_ALIAS=16 -DSWR_RASTER_MINWAVES=5" "-DSWR_VARY_GLOBAL=0 -DSWR_ABL_VARY_ALIAS=4 -DSWR_RASTER_MINWAVES=5" "-DSWR_VARY_GLOBAL=0 -DSWR_ABL_VARY_ALIAS=4 -DSWR_RASTER_MINWAVES=5 -DSWR_ABL_LDSBYTES=1024" "-DSWR_VARY_GLOBAL=0" > gpurun_out/ab_$TAG.txt 2>&1
python - <<P
import json,re
for ln in open('gpurun_out/ab_$TAG.txt'):
    m=re.match(r'(.*?)\s*(\{.*\})',ln)
    if m: d=json.loads(m.group(2)); print(m.group(1)[:100].ljust(100), d['raster_ms'], d['total_ms'])
    else: print(ln.rstrip()[:200])
P
