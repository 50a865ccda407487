#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_track; mkdir -p $OUT
timeout -k 5 300 python scripts/exp_track_run.py > $OUT/track_run_debug.txt 2>&1; grep -v amdgpu $OUT/track_run_debug.txt | tail -20
