#!/bin/bash
# round 5, session 5 (EXPERIMENTS build): the cell-run variants on sparse clouds (latency chain of demand loads): vectors per lane, run length, direct gather
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s5
V="runs,u3k4=D3F_EXP_RUNS_U=3,u3k2=D3F_EXP_RUNS_U=3+D3F_EXP_RUNS=2,u2=D3F_EXP_RUNS_U=2,k8=D3F_EXP_RUNS=8,direct=D3F_EXP_RUNS=-1,tile16=D3F_EXP_RUNS_TILE=16,tile64=D3F_EXP_RUNS_TILE=64"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s5 --variants "$V" \
  --cases c5_track:random,c3_patch:surface:r,ref_patch:surface:r,c2_patch:random,c4_patch:random 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s5/log.txt | grep -v '^{' | cut -c1-250
