#!/bin/bash
# round 4, session 43 (EXPERIMENTS build of the final sources): the whole GPU suite (incl. the tests that need tuning knobs)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4aj; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-200
