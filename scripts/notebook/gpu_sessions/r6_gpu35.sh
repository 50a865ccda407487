#!/bin/bash
# round 6, session 35: tests of the distance-only kernel on the PRODUCT build, then its SQ / vector-L1 counters (and the old branch's, experiments build)
set -u
REPO=$(pwd); TAG=${TAG:-r6_s35}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_dist.py -q -m gpu 2>&1 | tail -15 | cut -c1-250
TCP=1 bash scripts/r6_counters.sh $TAG dist_only 2>&1 | tail -2
cat $OUT/dist_only_sq.txt
