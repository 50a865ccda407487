#!/bin/bash
# round 3, session 31 (experiments build): C2-dense, 32- / 64-point workgroups of the sliced kernel at lower occupancy
set -u
export D3F_BUILD_EXPERIMENTS=1
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_c4sweep; mkdir -p $OUT
export TMPDIR=/tmp
S="D3F_EXP_SLICED=3,D3F_EXP_SLICED_VC=2"
EXP_REPS=2 timeout -k 5 900 python scripts/exp_knobs.py c2_dense "base:" "t32:$S,D3F_EXP_SLICED_TILE=32" "t32pad16:$S,D3F_EXP_SLICED_TILE=32,D3F_EXP_SLICED_PAD=16" "t32pad24:$S,D3F_EXP_SLICED_TILE=32,D3F_EXP_SLICED_PAD=24" "t32pad32:$S,D3F_EXP_SLICED_TILE=32,D3F_EXP_SLICED_PAD=32" "t32pad45:$S,D3F_EXP_SLICED_TILE=32,D3F_EXP_SLICED_PAD=45" "t64pad24:$S,D3F_EXP_SLICED_TILE=64,D3F_EXP_SLICED_PAD=24" "t64pad38:$S,D3F_EXP_SLICED_TILE=64,D3F_EXP_SLICED_PAD=38" "t64pad64:$S,D3F_EXP_SLICED_TILE=64,D3F_EXP_SLICED_PAD=64" > $OUT/c2_dense_sweep2.txt 2>&1
grep -v amdgpu $OUT/c2_dense_sweep2.txt | cut -c1-150
