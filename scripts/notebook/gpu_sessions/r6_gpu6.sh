#!/bin/bash
# round 6, session 6 (PRODUCT build, final sources): the default bench line with roofline.traffic measured in the run, the GPU suite, then
# the rocprofv3 evidence of every workload (scripts/r6_profile_all.sh -> gpurun_out/r6_v1)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s6; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python bench.py > $OUT/default_bench_line.json 2> $OUT/default_bench_line.err; tail -c 1500 $OUT/default_bench_line.json; tail -3 $OUT/default_bench_line.err
timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
scripts/r6_profile_all.sh r6_v1 > $OUT/profile_all.log 2>&1
tail -3 $OUT/profile_all.log
