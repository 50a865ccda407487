#!/bin/bash
# round 3, session 15: pipelined window kernel -- which side binds?  (debug 1: gather waves idle; 2: producer idle after two bricks)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3q; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
for WL in c2_patch c4_patch; do
timeout -k 5 600 python scripts/exp_knobs.py $WL "old window:D3F_EXP_WINPIPE=-1" "winpipe:" "winpipe P-only:D3F_EXP_STREAM_DEBUG=1" "winpipe C-only:D3F_EXP_STREAM_DEBUG=2" > $OUT/sweep_$WL.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep_$WL.txt | cut -c1-110
done
