#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_track; mkdir -p $OUT
timeout -k 5 300 python scripts/exp_track_run.py > $OUT/track_run_debug.txt 2>&1; grep -v amdgpu $OUT/track_run_debug.txt | cut -c1-110 | tail -12
timeout -k 5 600 python -m pytest tests -m gpu -q -x -k "rigid or track" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout -k 5 300 python scripts/exp_callers.py rigid > $OUT/rigid_timing.txt 2>&1; grep -v amdgpu $OUT/rigid_timing.txt | tail -12
