#!/bin/bash
# round 4, session 15 (EXPERIMENTS build): finer phase stamps of the window kernel's set-up
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4o; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
D3F_EXP_STAMPS=1 timeout -k 5 300 python scripts/exp_stamps.py c2_patch ref_patch > $OUT/stamps.txt 2>&1; grep -v amdgpu $OUT/stamps.txt | head -40
