#!/bin/bash
cd "$GRAFT_REPO_ROOT"; TAG=${1:-occ}
export ABLATE_N=30
python tools/ablate.py cfg3 lib:build_ab/r2.so "-DSWR_VARY_GLOBAL=0" "-DSWR_VARY_GLOBAL=0 -DSWR_BATCH=12 -DSWR_WINDOW=24 -DSWR_BATCH_FRAGS=1536" "-DSWR_VARY_GLOBAL=0 -DSWR_BATCH=12 -DSWR_WINDOW=32 -DSWR_BATCH_FRAGS=1536" "-DSWR_BATCH=13 -DSWR_WINDOW=26" "-DSWR_BATCH=13 -DSWR_WINDOW=32" "-DSWR_VARY_GLOBAL=0" lib:build_ab/r2.so > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
