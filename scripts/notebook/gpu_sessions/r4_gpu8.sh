#!/bin/bash
# round 4, session 8 (PRODUCT build): whole GPU suite, smoke(), every bench line after the window-kernel work
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4h; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log | cut -c1-200
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/r3_bench_all.sh r4h/bench | tail -16 | cut -c1-170
