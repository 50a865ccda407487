#!/bin/bash
# round 4, session 10 (PRODUCT build): rocprofv3 evidence of every single-GPU workload (kernel trace + PMC passes), the failed test again,
# dist_only bench line
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4j; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py -m gpu -q -k "c4_dense or dist_only" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-200
timeout -k 5 400 python bench.py --steps 20 --workload dist_only > $OUT/bench_dist_only.json 2> $OUT/bench_dist_only.err; tail -c 600 $OUT/bench_dist_only.json
bash scripts/r4_profile_all.sh r4_v1 | tail -12
du -sh $REPO/gpurun_out/r4_v1
