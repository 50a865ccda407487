#!/bin/bash
# round 3, session 21: correspondence kernels after the epilogue / tail-group / merge / vector-apply changes
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_corr; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q -x -k "pairwise or corr or softmax or knn or similarity or sharded or nearest" > $OUT/pytest_corr.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_corr.log
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c5_track c2_dense_f16 c2_patch_f16; do
  timeout -k 5 400 $B --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; tail -c 900 $OUT/bench_$WL.json; echo
done
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/c5_trace -o trace --output-format csv -- python $REPO/bench.py --workload c5_track --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $OUT/c5_under_rocprof.json 2> $OUT/c5_trace.err; cd $REPO
python scripts/kernel_stats.py $OUT/c5_trace d3f:: > $OUT/c5_kernel_stats.txt; head -12 $OUT/c5_kernel_stats.txt
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES -d $OUT/c5_pmc -o pmc --output-format csv -- python $REPO/bench.py --workload c5_track --steps 5 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/c5_pmc.err; cd $REPO
python - <<'PY' > $OUT/c5_counters.txt
import csv, glob, os
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for p in glob.glob("gpurun_out/r3_corr/c5_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "pairwise" in r["Kernel_Name"] or "softmax" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s n=%4d avg=%16.1f" % (c, len(v), sum(v) / len(v)))
PY
cat $OUT/c5_counters.txt
rm -rf $OUT/*/trace/*/*hip_api* 2>/dev/null; du -sh $OUT
