#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3i; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "thin_map_gather" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -12 $OUT/pytest.log
