#!/bin/bash
# round 3, session 29 (experiments build): C2-dense, four views in flight at lower occupancy (fewer points in flight per XCD)
set -u
export D3F_BUILD_EXPERIMENTS=1
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_c4sweep; mkdir -p $OUT
export TMPDIR=/tmp
S="D3F_EXP_SLICED=3"
EXP_REPS=2 timeout -k 5 900 python scripts/exp_knobs.py c2_dense "base:" "vc4:$S,D3F_EXP_SLICED_VC=4" "vc4pad28:$S,D3F_EXP_SLICED_VC=4,D3F_EXP_SLICED_PAD=28" "vc4pad36:$S,D3F_EXP_SLICED_VC=4,D3F_EXP_SLICED_PAD=36" "vc4pad48:$S,D3F_EXP_SLICED_VC=4,D3F_EXP_SLICED_PAD=48" "vc2pad20:$S,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_PAD=20" "vc2pad24:$S,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_PAD=24" "vc4t8:$S,D3F_EXP_SLICED_VC=4,D3F_EXP_SLICED_TILE=8" "vc4t32:$S,D3F_EXP_SLICED_VC=4,D3F_EXP_SLICED_TILE=32" "vc4t32pad40:$S,D3F_EXP_SLICED_VC=4,D3F_EXP_SLICED_TILE=32,D3F_EXP_SLICED_PAD=40" > $OUT/c2_dense_sweep.txt 2>&1
grep -v amdgpu $OUT/c2_dense_sweep.txt | cut -c1-170
