#!/bin/bash
# round 4, session 1: folded weights (4 fma per view and vector) + device-side finite words: whole GPU suite (all failures),
# then every bench line
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4a; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest_gpu.log | cut -c1-220
bash scripts/r3_bench_all.sh r4a/bench | tail -16 | cut -c1-170
