#!/bin/bash
# round 5, session 3 (EXPERIMENTS build): LDS slack of the sparse window kernel at 3 workgroups per CU; Hilbert vs Morton again, two rounds per variant
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s3
V="runs,runs_morton=D3F_EXP_ORDER_MORTON=1,win_sparse=D3F_EXP_WINDOW=64,win_sparse_s2560=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SLACK=2560,win_sparse_s6144=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SLACK=6144,win_rect=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SPARSE=-1+D3F_EXP_WINDOW_SLACK=2048,win_sparse_w10=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_WANT=10"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s3 --variants "$V" \
  --cases c2_patch:random,c3_patch:random,c5_track:random,c2_patch:grid,c3_patch:grid 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s3/log.txt | grep -v '^{' | cut -c1-250
