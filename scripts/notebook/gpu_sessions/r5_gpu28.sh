#!/bin/bash
# round 5, session 28 (EXPERIMENTS build): C4-patch cloud (8 views, 1024-d) on the window kernel with 32-point tiles (their touched texels fit the pool more often)
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s28
V="auto,w32=D3F_EXP_WINDOW=32+D3F_EXP_GATE=1,w64=D3F_EXP_WINDOW=64+D3F_EXP_GATE=1,w32gate=D3F_EXP_WINDOW=32"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s28 --variants "$V" --steps 20 --cases c4_patch:random,ref_patch:random 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s28/log.txt | grep -v '^{' | cut -c1-200
