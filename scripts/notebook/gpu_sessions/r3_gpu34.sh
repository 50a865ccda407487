#!/bin/bash
# round 3, session 34 (experiments build): C4-patch lattice, brick size / occupancy / views in flight / slice width of the window kernel
set -u
export D3F_BUILD_EXPERIMENTS=1
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_c4sweep; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python scripts/exp_knobs.py c4_patch "base:" "w32:D3F_EXP_WINDOW=32" "w32o3:D3F_EXP_WINDOW=32,D3F_EXP_WINDOW_OCC=3" "w32o2:D3F_EXP_WINDOW=32,D3F_EXP_WINDOW_OCC=2" "w64o3:D3F_EXP_WINDOW=64,D3F_EXP_WINDOW_OCC=3" "w64vc2:D3F_EXP_WINDOW=64,D3F_EXP_WINDOW_VC=2" "w64u2:D3F_EXP_WINDOW=64,D3F_EXP_WINDOW_U=2" "w64lpp32:D3F_EXP_WINDOW=64,D3F_EXP_WINDOW_LPP=32" "w128:D3F_EXP_WINDOW=128" "w32vc2:D3F_EXP_WINDOW=32,D3F_EXP_WINDOW_VC=2" "base2:" > $OUT/c4_patch_window_sweep.txt 2>&1
grep -v amdgpu $OUT/c4_patch_window_sweep.txt | cut -c1-160
