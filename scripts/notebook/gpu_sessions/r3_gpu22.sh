#!/bin/bash
# round 3, session 22: 1024-lane statistics merge, integer probe positions; channel-sliced launch for C = 1024 (one 512-byte
# slice per XCD) on the C4-dense lattice slab
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_c4s; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q -x -k "pairwise or corr or softmax or knn or similarity or sharded or nearest or sliced or c4_dense or plan or order or probe" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c5_track c4_dense; do
  timeout -k 5 600 $B --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  python - $OUT/bench_$WL.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print(d["config"]["workload"][:40], "| step %.3f ms | kernel %.3f ms | frac %.3f | traffic %s | verified %s | %s | %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r.get("traffic"), d.get("verified"), r["kernel"], d["config"].get("point_order")))
PY
done
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/c5_trace -o trace --output-format csv -- python $REPO/bench.py --workload c5_track --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $OUT/c5_under_rocprof.json 2> $OUT/c5_trace.err; cd $REPO
python scripts/kernel_stats.py $OUT/c5_trace d3f:: > $OUT/c5_kernel_stats.txt; head -12 $OUT/c5_kernel_stats.txt
cd /tmp
for PMC in "FETCH_SIZE WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  N=$(echo $PMC | tr ' ' '_' | cut -c1-20)
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/c4_pmc_$N -o pmc --output-format csv -- python $REPO/bench.py --workload c4_dense --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/c4_pmc_$N.err
done
cd $REPO
python - <<'PY' > $OUT/c4_counters.txt
import csv, glob, os
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for p in glob.glob("gpurun_out/r3_c4s/c4_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "fused_eval" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s n=%4d avg=%16.1f" % (c, len(v), sum(v) / len(v)))
PY
cat $OUT/c4_counters.txt
rm -rf $OUT/*/trace/*/*hip_api* 2>/dev/null; du -sh $OUT
