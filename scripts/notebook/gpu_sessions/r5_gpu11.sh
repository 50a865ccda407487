#!/bin/bash
# round 5, session 11 (EXPERIMENTS build): the ordering's scan -- one launch with decoupled look-back vs the three-launch scan, same box
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s11
V="lookback,scan3=D3F_EXP_SCAN3=1"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s11 --variants "$V" --steps 40 \
  --cases c5_track:random,c2_patch:random 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s11/log.txt | grep -v '^{' | cut -c1-250
REPO=$(pwd); cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r5_s11/c5/trace -o trace --output-format csv -- python $REPO/bench.py --workload c5_track --no-cpu-baseline --no-verify --steps 30 > $REPO/gpurun_out/r5_s11/c5_bench.json 2> $REPO/gpurun_out/r5_s11/c5.trace.err
cd $REPO; python scripts/summarize_prof.py gpurun_out/r5_s11/c5 > gpurun_out/r5_s11/c5_track_summary.txt 2>&1; rm -rf gpurun_out/r5_s11/c5/trace/*/*hip_api*; grep -E "scan_|pairwise|runs_kernel" gpurun_out/r5_s11/c5_track_summary.txt | head -5 | cut -c1-160
