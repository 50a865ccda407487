#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2e; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -6 $OUT/pytest.log
timeout -k 5 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
bash scripts/r2_profile_all.sh r2_v1 > $OUT/profile_all.log 2>&1; tail -3 $OUT/profile_all.log
for f in gpurun_out/r2_v1/*_summary.txt; do echo "== $f"; grep -A3 "per-kernel durations" $f | head -5; grep "fused_eval\|pairwise" $f | grep "avg=" | head -12; done
