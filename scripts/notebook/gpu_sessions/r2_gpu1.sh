#!/bin/bash
# Round 2, GPU call 1: GPU test suite, bench lines of every workload with the new paths on/off, DRAM-side counters.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2a; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline"
echo "== tests" ; date
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
echo "== bench lines"; date
for WL in c2_dense c3_dense c2_patch c3_patch c4_patch; do
  timeout -k 5 200 $B --workload $WL --steps 20 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
done
# the old paths for comparison: Morton sort instead of the lattice walk; direct gather instead of cell runs
D3F_EXP_WALK=-1 timeout -k 5 200 $B --workload c2_dense --steps 20 --no-verify > $OUT/bench_c2_dense_sort.json 2>&1
D3F_EXP_WALK=-1 timeout -k 5 200 $B --workload c3_dense --steps 20 --no-verify > $OUT/bench_c3_dense_sort.json 2>&1
for WL in c2_patch c3_patch c4_patch; do
  D3F_EXP_RUNS=-1 timeout -k 5 200 $B --workload $WL --steps 20 --no-verify > $OUT/bench_${WL}_direct.json 2>&1
  D3F_EXP_RUNS=4 timeout -k 5 200 $B --workload $WL --steps 20 --no-verify > $OUT/bench_${WL}_runs4.json 2>&1
  D3F_EXP_RUNS=8 D3F_EXP_RUNS_OCC=5 timeout -k 5 200 $B --workload $WL --steps 20 --no-verify > $OUT/bench_${WL}_runs8occ5.json 2>&1
done
timeout -k 5 200 $B --workload c2_patch --points random --steps 20 --no-verify > $OUT/bench_c2_patch_random.json 2>&1
D3F_EXP_RUNS=-1 timeout -k 5 200 $B --workload c2_patch --points random --steps 20 --no-verify > $OUT/bench_c2_patch_random_direct.json 2>&1
D3F_EXP_RUNS=-1 timeout -k 5 200 $B --workload c4_patch --tuning 0x8000 --steps 20 --no-verify > $OUT/bench_c4_patch_staged.json 2>&1
echo "== counters"; date
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -o "TCC_EA0_RD[A-Za-z0-9_]*\|TCC_EA0_WR[A-Za-z0-9_]*\|TCC_[A-Z_]*MALL[A-Za-z0-9_]*\|TCC_BUBBLE[a-z_]*\|TCP_TCC_READ_REQ[A-Za-z_]*" $OUT/counters_list.txt | sort -u > $OUT/counters_tcc.txt
cd /tmp
CMD="$B --workload c2_dense --steps 5 --warmup 1 --no-verify"
i=0
for PMC in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc_c2_dense/p$i -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_c2_dense_p$i.err
done
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/trace_default -o trace --output-format csv -- python $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_default_under_rocprof.json 2> $OUT/trace_default.err
cd $REPO
python - "$OUT" <<'PY' > $OUT/pmc_c2_dense_summary.txt 2>&1
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for p in glob.glob(os.path.join(root, "pmc_c2_dense", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "fused_eval" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-40s %18.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUT/pmc_c2_dense_summary.txt
python scripts/summarize_prof.py $OUT/trace_default > $OUT/trace_default_summary.txt 2>&1
head -30 $OUT/trace_default_summary.txt
for f in $OUT/bench_*.json; do echo "$(basename $f): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    print("step %.3f ms kernel %.3f ms frac %.3f value %.3e verified %s %s" % (d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["value"], d.get("verified"), d["roofline"]["kernel"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
date
