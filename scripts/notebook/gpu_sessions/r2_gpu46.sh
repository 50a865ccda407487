#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3q; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
