#!/bin/bash
# round 3, session 44: final state -- whole GPU suite, smoke(), every bench line, default bench under the kernel trace
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_final; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/r3_bench_all.sh r3_final/bench | tail -16 | cut -c1-150
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/default_trace -o trace --output-format csv -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/default_bench_under_rocprof.json 2> $OUT/default_trace.err; cd $REPO
python scripts/kernel_stats.py $OUT/default_trace d3f:: > $OUT/default_bench_kernel_stats.txt; head -4 $OUT/default_bench_kernel_stats.txt
timeout -k 5 300 python scripts/exp_callers.py rigid fps > $OUT/callers_timing.txt 2>&1; grep -v amdgpu $OUT/callers_timing.txt | tail -16 | cut -c1-150
rm -rf $OUT/*/trace/*/*hip_api* 2>/dev/null; du -sh $OUT
