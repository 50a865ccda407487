#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2i; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -6 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
bash scripts/r2_bench_all.sh r2_bench2 > $OUT/bench_all.log 2>&1; cat $OUT/bench_all.log
bash scripts/r2_profile_all.sh r2_v2 c2_dense c3_dense c2_patch c3_patch c4_patch c5_track c4_dense > $OUT/profile_all.log 2>&1; tail -3 $OUT/profile_all.log
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_default -o trace --output-format csv -- python $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_default_under_rocprof.json 2> $OUT/trace_default.err
cd $REPO
python scripts/summarize_prof.py $OUT/trace_default > $OUT/default_bench_kernel_stats.txt 2>&1; head -12 $OUT/default_bench_kernel_stats.txt
