#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_c4sweep; mkdir -p $OUT
timeout -k 5 300 python scripts/exp_c3_mask.py > $OUT/c3_mask_cost.txt 2>&1; grep -v amdgpu $OUT/c3_mask_cost.txt
