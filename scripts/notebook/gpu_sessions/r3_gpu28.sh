#!/bin/bash
# round 3, session 28 (experiments build): 8-point tiles of the channel-sliced kernel on C4-dense, with occupancy / views in flight
set -u
export D3F_BUILD_EXPERIMENTS=1
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_c4sweep; mkdir -p $OUT
export TMPDIR=/tmp
S="D3F_EXP_SLICED=3,D3F_EXP_SLICED_TILE=8"
timeout -k 5 900 python scripts/exp_knobs.py c4_dense "base:" "t8:$S,D3F_EXP_SLICED_VC=2" "t8vc1:$S,D3F_EXP_SLICED_VC=1" "t8vc4:$S,D3F_EXP_SLICED_VC=4" "t8pad20:$S,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_PAD=20" "t8pad26:$S,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_PAD=26" "t8pad34:$S,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_PAD=34" "t8u256:$S,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_UNIT=256" "t8u1024:$S,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_UNIT=1024" "t8vc1pad20:$S,D3F_EXP_SLICED_VC=1,D3F_EXP_SLICED_PAD=20" "base2:" > $OUT/c4_dense_sweep2.txt 2>&1
grep -v amdgpu $OUT/c4_dense_sweep2.txt | cut -c1-170
timeout -k 5 900 python scripts/exp_knobs.py c3_dense "base:" "t16:D3F_EXP_SLICED=3,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_TILE=16" "t8:D3F_EXP_SLICED=3,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_TILE=8" "pad8:D3F_EXP_SLICED=3,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_PAD=8" "pad14:D3F_EXP_SLICED=3,D3F_EXP_SLICED_VC=2,D3F_EXP_SLICED_PAD=14" "base2:" > $OUT/c3_dense_sweep.txt 2>&1
grep -v amdgpu $OUT/c3_dense_sweep.txt | cut -c1-170
