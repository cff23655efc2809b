#!/bin/bash
cd "$GRAFT_REPO_ROOT"; TAG=${1:-vg3}; export TMPDIR=/tmp
export ABLATE_N=30
python tools/ablate.py cfg3 lib:build_ab/r2.so "" "-DSWR_BATCH=14 -DSWR_WINDOW=28" "-DSWR_BATCH=12 -DSWR_WINDOW=24" "-DSWR_BATCH=8 -DSWR_WINDOW=16 -DSWR_RASTER_MINWAVES=6 -DSWR_BATCH_FRAGS=1024" "-DSWR_VARY_GLOBAL=0" > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
bash tools/pmc_run.sh $TAG cfg3 > gpurun_out/pmc_${TAG}_run.log 2>&1; tail -30 gpurun_out/pmc_${TAG}.txt
