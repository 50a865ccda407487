#!/bin/bash
# end-of-round evidence, third pass: profiles of the workloads whose kernel changed + default bench under rocprof
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2_final3; mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/r2_profile_all.sh r2_v3 c2_patch c3_patch > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r2_v3/default_bench -o trace --output-format csv -- python $REPO/bench.py > $REPO/gpurun_out/r2_v3/default_bench_under_rocprof.json 2> $OUT/default_bench.err
cd $REPO
python scripts/summarize_prof.py gpurun_out/r2_v3/default_bench > gpurun_out/r2_v3/default_bench_kernel_stats.txt 2>&1
tail -1 gpurun_out/r2_v3/default_bench_under_rocprof.json | cut -c1-300
rm -rf gpurun_out/r2_v3/*/trace/*/*hip_api* gpurun_out/r2_v3/default_bench/*/*hip_api* 2>/dev/null
