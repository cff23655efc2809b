#!/bin/bash
# k_cover: what the shape sort leaves on the table (useful / executed pixel steps), and block size / sort key variants timed
cd "$(dirname "$0")/../.." || exit 1
T=${1:-cover}
O=gpurun_out/ab_$T.txt; : > $O
for v in "" "-DSWR_COVER_BLOCK=512" "-DSWR_COVER_BLOCK=1024" "-DSWR_COVER_HQ=2 -DSWR_COVER_WQ=2" "-DSWR_COVER_BLOCK=1024 -DSWR_COVER_HQ=2 -DSWR_COVER_WQ=2" "-DSWR_COVER_HQ=1 -DSWR_COVER_WQ=4"; do
  python3 tools/debug_counters.py --cover cfg3 $v >> $O 2>&1 || exit 1
done
ABLATE_N=30 python3 tools/ablate.py cfg3 "" "-DSWR_COVER_BLOCK=512" "-DSWR_COVER_BLOCK=1024" "-DSWR_COVER_HQ=2 -DSWR_COVER_WQ=2" "-DSWR_COVER_BLOCK=512 -DSWR_COVER_HQ=2 -DSWR_COVER_WQ=2" "-DSWR_COVER_BLOCK=1024 -DSWR_COVER_HQ=2 -DSWR_COVER_WQ=2" "-DSWR_COVER_HQ=1 -DSWR_COVER_WQ=4" "-DSWR_COVER_BLOCK=1024 -DSWR_COVER_HQ=1 -DSWR_COVER_WQ=2" "" >> $O 2>&1
cat $O
