#!/bin/bash
# round 4, session 32 (PRODUCT build): the remaining workloads' rocprofv3 summaries + counter passes with the final kernels
# (c4_dense, c5_track, dist_only -> profiles/r4_v3/), and the default bench line once more (traffic now from profiles/r4_v3)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_v3; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python $REPO/bench.py > $OUT/default_bench.json 2> $OUT/default_bench.err; tail -c 600 $OUT/default_bench.json | head -c 300; echo
bash $REPO/scripts/r4_profile_all.sh r4_v3 c4_dense c5_track dist_only > $OUT/profile_all2.log 2>&1
rm -rf $OUT/*/trace $OUT/*/pmc_* 2>/dev/null
ls $OUT/*_summary.txt
