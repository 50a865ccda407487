#!/bin/bash
# round 4, session 29 (PRODUCT build): stability of the non-temporal rows -- the window / store / parity tests twelve times over
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ab; mkdir -p $OUT
export TMPDIR=/tmp
for I in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout -k 5 600 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x -k "window or stores or fast_path or strict or golden" > $OUT/run_$I.log 2>&1
  echo "run $I rc=$? $(tail -1 $OUT/run_$I.log | cut -c1-100)"
done
