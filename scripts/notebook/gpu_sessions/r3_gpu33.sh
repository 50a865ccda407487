#!/bin/bash
# round 3, session 33 (experiments build): LDS texel windows forced on the Morton-ordered C4-patch / C2-patch / C3-patch CLOUDS
set -u
export D3F_BUILD_EXPERIMENTS=1
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_c4sweep; mkdir -p $OUT
export TMPDIR=/tmp
for WL in c4_patch c2_patch c3_patch; do
timeout -k 5 600 python scripts/exp_knobs.py $WL:random "base:" "win64:D3F_EXP_WINDOW=64" "win32:D3F_EXP_WINDOW=32" "win64occ2:D3F_EXP_WINDOW=64,D3F_EXP_WINDOW_OCC=2" "win64occ3:D3F_EXP_WINDOW=64,D3F_EXP_WINDOW_OCC=3" "win128:D3F_EXP_WINDOW=128" > $OUT/${WL}_cloud_window.txt 2>&1
grep -v amdgpu $OUT/${WL}_cloud_window.txt | cut -c1-160
done
