#!/bin/bash
# round 3, session 16: pipelined window kernel, 4 / 8 / 12 gather waves per workgroup
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3r; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
N="D3F_EXP_WINPIPE_NG"
for WL in c2_patch c3_patch c4_patch; do
timeout -k 5 600 python scripts/exp_knobs.py $WL "old window:D3F_EXP_WINPIPE=-1" "ng4:$N=4" "ng8:$N=8" "ng12:$N=12" "ng8 occ1:$N=8,D3F_EXP_WINPIPE_OCC=1" "ng12 occ1:$N=12,D3F_EXP_WINPIPE_OCC=1" "ng8 G2:$N=8,D3F_EXP_WINPIPE_G=2" "ng8 C-only:$N=8,D3F_EXP_STREAM_DEBUG=2" "old window again:D3F_EXP_WINPIPE=-1" > $OUT/sweep_$WL.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep_$WL.txt | cut -c1-120
done
