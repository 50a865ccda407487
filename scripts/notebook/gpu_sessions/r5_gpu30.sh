#!/bin/bash
# round 5, session 30 (EXPERIMENTS build + scripts/notebook/patches/r5_s30_window_occ1_knob.patch): C4-patch cloud on the window kernel with ONE workgroup per CU (the whole LDS as one 272-slot pool: every 64-point tile's ~176 touched texels fit)
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s30
V="auto,occ1=D3F_EXP_WINDOW_OCC=1,occ1forced=D3F_EXP_WINDOW_OCC=1+D3F_EXP_GATE=1"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s30 --variants "$V" --steps 20 --cases c4_patch:random,c4_patch:grid 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s30/log.txt | grep -v '^{' | cut -c1-200
grep oracle gpurun_out/r5_s30/log.txt | cut -c1-200
