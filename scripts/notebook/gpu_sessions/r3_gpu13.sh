#!/bin/bash
# round 3, session 13: window kernel on config 4's lattice slab: pool size (workgroups per CU), brick size
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3o; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
W="D3F_EXP_WINDOW"
timeout -k 5 600 python scripts/exp_knobs.py c4_patch "default:" "occ3:$W=64,${W}_OCC=3" "occ2:$W=64,${W}_OCC=2" "T32:$W=32" "T32occ3:$W=32,${W}_OCC=3" "T128occ2:$W=128,${W}_OCC=2" "lpp32:$W=64,${W}_LPP=32" "lpp32occ2:$W=64,${W}_LPP=32,${W}_OCC=2" "U2:$W=64,${W}_U=2" "runs:$W=-1" "direct:$W=-1,D3F_EXP_RUNS=-1" > $OUT/sweep_c4.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep_c4.txt | cut -c1-150
timeout -k 5 600 python scripts/exp_knobs.py c2_patch "default:" "occ3:$W=64,${W}_OCC=3" "occ2:$W=64,${W}_OCC=2" "T32:$W=32" "T128occ2:$W=128,${W}_OCC=2" > $OUT/sweep_c2.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep_c2.txt | cut -c1-150
