#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2r; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python scripts/exp_window.py c2_patch,c3_patch > $OUT/window.log 2>&1; echo "rc=$?" >> $OUT/window.log
cat $OUT/window.log | tail -40
