#!/bin/bash
# round 3, session 24: all optimiser steps of a tracking frame in one launch (d3f_track_run)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_track; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests -m gpu -q -x -k "rigid or track" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout -k 5 300 python scripts/exp_callers.py rigid > $OUT/rigid_timing.txt 2>&1; grep -v amdgpu $OUT/rigid_timing.txt | tail -14
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $REPO/scripts/exp_callers.py rigid > /dev/null 2> $OUT/trace.err; cd $REPO
python scripts/kernel_stats.py $OUT/trace d3f:: > $OUT/rigid_kernel_stats.txt; head -8 $OUT/rigid_kernel_stats.txt
rm -rf $OUT/*/trace/*/*hip_api* 2>/dev/null; du -sh $OUT
