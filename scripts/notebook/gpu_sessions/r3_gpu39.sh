#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_track; mkdir -p $OUT
timeout -k 5 200 python scripts/exp_track_run.py > $OUT/track_run_instrumented.txt 2>&1; grep -E "^wave|^updater" $OUT/track_run_instrumented.txt | head -40; grep -E "run-eager" $OUT/track_run_instrumented.txt | cut -c1-80
