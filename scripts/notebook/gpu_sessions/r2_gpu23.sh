#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2w; mkdir -p $OUT
export TMPDIR=/tmp
python scripts/dbg_window.py 2>&1 | grep -v amdgpu.ids | grep "U2" | cut -c1-140
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
bash scripts/r2_bench_all.sh r2w 2>&1 | tail -14
