#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3c; mkdir -p $OUT
export TMPDIR=/tmp
export D3F_EXP_WINDOW=64 D3F_EXP_WINDOW_OCC=3
bash scripts/pmc_any.sh win64o3 fused_eval_window python $REPO/bench.py --no-cpu-baseline --no-verify --steps 3 --warmup 1 --workload c2_patch > $OUT/pmc_win64o3.txt 2>&1
cat $OUT/pmc_win64o3.txt
