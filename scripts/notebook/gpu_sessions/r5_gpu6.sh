#!/bin/bash
# round 5, session 6 (EXPERIMENTS build): the device-side gate between the sparse window kernel and the cell runs on clouds
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s6
V="auto,runs_only=D3F_EXP_GATE=-1,window_only=D3F_EXP_GATE=1,old_runs=D3F_EXP_WINDOW=-1"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s6 --variants "$V" \
  --cases c2_patch:random,c3_patch:random,ref_patch:random,c5_track:random,ref_patch:surface:r,c4_patch:random,c2_patch:grid 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s6/log.txt | grep -v '^{' | cut -c1-250
grep oracle gpurun_out/r5_s6/log.txt | cut -c1-200
