#!/bin/bash
# round 5, session 1 (EXPERIMENTS build): Hilbert vs Morton order, cell runs vs LDS windows on clouds (same box, one process)
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_cloud
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_cloud 2>&1 | grep -v amdgpu | tee gpurun_out/r5_cloud/log.txt | tail -80 | cut -c1-330
