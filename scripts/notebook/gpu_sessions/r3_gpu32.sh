#!/bin/bash
# round 3, session 32: whole GPU suite, smoke(), every bench line, rocprofv3 summaries of the workloads whose kernels changed (r3_v2)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_v2; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/r3_bench_all.sh r3_v2/bench | tail -16
bash scripts/r3_profile_all.sh r3_v2 c2_dense c4_dense c5_track > /dev/null 2>&1; ls $OUT/*_summary.txt | wc -l
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/default_trace -o trace --output-format csv -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/default_bench_under_rocprof.json 2> $OUT/default_trace.err; cd $REPO
python scripts/kernel_stats.py $OUT/default_trace d3f:: > $OUT/default_bench_kernel_stats.txt; head -5 $OUT/default_bench_kernel_stats.txt
timeout -k 5 300 python bench.py > $OUT/default_bench.json 2> $OUT/default_bench.err; tail -c 600 $OUT/default_bench.json
rm -rf $OUT/*/trace/*/*hip_api* $OUT/*/*/*/*hip_api* 2>/dev/null; du -sh $OUT
