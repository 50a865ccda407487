#!/bin/bash
# round 3, session 14: the pipelined window kernel (producer wave + gather waves) vs the round-2 window kernel
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3p; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
for WL in c2_patch c3_patch c4_patch; do
timeout -k 5 600 python scripts/exp_knobs.py $WL "direct:D3F_EXP_WINDOW=-1,D3F_EXP_RUNS=-1" "old window:D3F_EXP_WINPIPE=-1" "winpipe:" "winpipe G2:D3F_EXP_WINPIPE_G=2" "winpipe occ1:D3F_EXP_WINPIPE_OCC=1" "winpipe occ1 G2:D3F_EXP_WINPIPE_OCC=1,D3F_EXP_WINPIPE_G=2" "old window again:D3F_EXP_WINPIPE=-1" > $OUT/sweep_$WL.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep_$WL.txt | cut -c1-170
done
