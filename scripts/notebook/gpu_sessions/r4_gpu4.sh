#!/bin/bash
# round 4, session 4 (EXPERIMENTS build): phase stamps of the window kernel on the three patch workloads (pipelined and plain point loop)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4d; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
D3F_EXP_STAMPS=1 timeout -k 5 300 python scripts/exp_stamps.py c2_patch c3_patch c4_patch > $OUT/stamps_pipelined.txt 2>&1; grep -v amdgpu $OUT/stamps_pipelined.txt
D3F_EXP_WINDOW_PIPE=-1 D3F_EXP_STAMPS=1 timeout -k 5 300 python scripts/exp_stamps.py c2_patch c4_patch > $OUT/stamps_plain.txt 2>&1; grep -v amdgpu $OUT/stamps_plain.txt
