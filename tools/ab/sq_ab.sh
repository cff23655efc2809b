#!/bin/bash
# SQ counters of k_raster_c for the product and the rejected decompositions of round 3, one box, one call:
# for each variant the stage times (tools/ablate.py) and the two SQ passes (tools/pmc_run.sh) -> gpurun_out/sq_ab.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/sq_ab.txt; : > $O
i=0
for v in "" "-DSWR_VARY_GLOBAL=1" "-DSWR_BATCH=12 -DSWR_WINDOW=24 -DSWR_BATCH_FRAGS=1536" "-DSWR_BATCH=8 -DSWR_WINDOW=16 -DSWR_RASTER_MINWAVES=6 -DSWR_BATCH_FRAGS=1024" "-DSWR_ABL_LDSBYTES=2048" "-DSWR_BATCH=24 -DSWR_WINDOW=48 -DSWR_VARY_GLOBAL=1"; do
  echo "=== variant $i: ${v:-product}" >> $O
  ABLATE_N=30 python3 tools/ablate.py cfg3 "$v" >> $O 2>&1 || exit 1
  make -C softwarerenderer_amd/csrc -s -B EXTRA="$v" 2>/dev/null || exit 1
  bash tools/pmc_run.sh sqab$i >> $O 2>&1 || exit 1
  i=$((i+1))
done
make -C softwarerenderer_amd/csrc -s -B 2>/dev/null
grep -c "SQ_" $O
