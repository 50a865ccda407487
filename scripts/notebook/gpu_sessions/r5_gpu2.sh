#!/bin/bash
# round 5, session 2 (EXPERIMENTS build): touched-texel pool (SPARSE) vs whole rectangles in the window kernel, clouds and lattices
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_sparse
V="runs,win_sparse=D3F_EXP_WINDOW=64,win_rect=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SPARSE=-1,win_sparse_w10=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_WANT=10,win_sparse_w8=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_WANT=8,win_rect_w10=D3F_EXP_WINDOW=64+D3F_EXP_WINDOW_SPARSE=-1+D3F_EXP_WINDOW_WANT=10"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_sparse --variants "$V" \
  --cases c2_patch:random,c3_patch:random,ref_patch:random,ref_patch:surface:r,c2_patch:grid,c3_patch:grid,ref_patch:grid,c4_patch:grid 2>&1 | grep -v amdgpu | tee gpurun_out/r5_sparse/log.txt | grep -v '^{' | cut -c1-250
