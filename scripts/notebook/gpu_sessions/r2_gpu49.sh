#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3s; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python scripts/exp_rigid.py 2>&1 | grep -v amdgpu.ids | tee $OUT/rigid.log
timeout -k 5 900 python -m pytest tests -m gpu -q -x -k "rigid or track or driver" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
