#!/bin/bash
# end-of-round evidence: full GPU suite, smoke, rocprofv3 summaries of every workload, every bench line, default bench under rocprof
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2_final; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
bash scripts/r2_profile_all.sh r2_v3 c2_dense c3_dense c2_patch c3_patch c4_patch c5_track > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
bash scripts/r2_bench_all.sh r2_v3_bench 2>&1 | tail -14
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r2_v3/default_bench -o trace --output-format csv -- python $REPO/bench.py > $REPO/gpurun_out/r2_v3/default_bench_under_rocprof.json 2> $OUT/default_bench.err
cd $REPO
python scripts/summarize_prof.py gpurun_out/r2_v3/default_bench > gpurun_out/r2_v3/default_bench_kernel_stats.txt 2>&1
tail -1 gpurun_out/r2_v3/default_bench_under_rocprof.json | cut -c1-400
rm -rf gpurun_out/r2_v3/*/trace/*/*hip_api* gpurun_out/r2_v3/default_bench/*/*hip_api* 2>/dev/null
du -sh gpurun_out/r2_v3 | tail -1
