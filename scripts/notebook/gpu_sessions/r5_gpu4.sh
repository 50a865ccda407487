#!/bin/bash
# round 5, session 4 (EXPERIMENTS build): why is the sparse window kernel slow at 3 workgroups per CU when thin maps ride along?  LDS slack sweep
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s4
V="win_rect=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SPARSE=-1+D3F_EXP_WINDOW_SLACK=2048,sp4096=D3F_EXP_WINDOW=64,sp8192=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SLACK=8192,sp12288=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SLACK=12288,sp20000=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SLACK=20000,rect8192=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SPARSE=-1+D3F_EXP_WINDOW_SLACK=8192,rect12288=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SPARSE=-1+D3F_EXP_WINDOW_SLACK=12288"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s4 --variants "$V" \
  --cases c3_patch:grid,c2_patch:grid 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s4/log.txt | grep -v '^{' | cut -c1-250
