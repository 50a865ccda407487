#!/bin/bash
# thin-map view-parallel gather: full GPU suite, then every workload's bench line
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2q; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
bash scripts/r2_bench_all.sh r2q
