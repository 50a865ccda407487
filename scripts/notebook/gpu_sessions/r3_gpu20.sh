#!/bin/bash
# round 3, session 20: the whole GPU suite, every bench line, rocprofv3 evidence per workload and for the caller-side kernels
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_v1; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
bash scripts/r3_bench_all.sh r3_v1/bench | tail -16
bash scripts/r3_profile_all.sh r3_v1 > /dev/null 2>&1; ls $OUT/*_summary.txt | wc -l
# the default bench command under the kernel trace (what the driver runs), and the caller-side kernels
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/default_trace -o trace --output-format csv -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/default_bench_under_rocprof.json 2> $OUT/default_trace.err; cd $REPO
python scripts/kernel_stats.py $OUT/default_trace d3f:: > $OUT/default_bench_kernel_stats.txt; head -5 $OUT/default_bench_kernel_stats.txt
cd /tmp; timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT/callers_trace -o callers --output-format csv -- python $REPO/scripts/exp_callers.py > $OUT/callers_timing_under_rocprof.txt 2> $OUT/callers_trace.err
for PMC in "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $PMC | tr ' ' '_' | cut -c1-20)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/callers_pmc_$N -o pmc --output-format csv -- python $REPO/scripts/exp_callers.py dist shell backward > /dev/null 2> $OUT/callers_pmc_$N.err
done
cd $REPO
python scripts/kernel_stats.py $OUT/callers_trace d3f:: > $OUT/callers_kernel_stats.txt
python - <<'PY' > $OUT/callers_counters.txt
import csv, glob, os
from collections import defaultdict
root = "gpurun_out/r3_v1"
agg = defaultdict(lambda: defaultdict(list))
for p in glob.glob(os.path.join(root, "callers_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "d3f::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s n=%4d avg=%16.1f max=%16.1f" % (c, len(v), sum(v) / len(v), max(v)))
PY
python scripts/exp_callers.py > $OUT/callers_timing.txt 2>&1; grep -v amdgpu $OUT/callers_timing.txt
rm -rf $OUT/*/trace/*/*hip_api* 2>/dev/null; du -sh $OUT
