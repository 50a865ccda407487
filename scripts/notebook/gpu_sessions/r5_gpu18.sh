#!/bin/bash
# round 5, session 18 (EXPERIMENTS what-if builds -DD3F_RUNS_PREFETCH): the cell-run gather with the next point's new cell touched one step ahead
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r5_s18; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
for AB in 1 0 1; do
  cp $REPO/build_ab/runs_$AB.so $REPO/d3fields_amd/libd3fields_hip.so
  timeout -k 5 600 python scripts/notebook/exp_cloud.py --out $OUT --variants "runs=D3F_EXP_GATE=-1" --steps 30 --cases c5_track:random,c4_patch:random,ref_patch:surface,c2_patch:random > $OUT/log_$AB.txt 2>&1
  grep -E "^c[0-9]_|^ref_|Error|error" $OUT/log_$AB.txt | awk -v ab=$AB '{print "prefetch", ab, $0}' | cut -c1-170
  tail -3 $OUT/log_$AB.txt | cut -c1-300
done
