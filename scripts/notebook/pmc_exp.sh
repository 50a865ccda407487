#!/bin/bash
# scripts/pmc_exp.sh <tag> <kernel-name substring> <workload> <variant spec for exp_knobs.py>
# two counter passes (memory side, SQ side) of ONE knob variant of one workload; prints per-dispatch averages
set -u
TAG=$1; KERNEL=$2; WL=$3; VAR=$4
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for PMC in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o pmc --output-format csv -- python $REPO/scripts/exp_knobs.py $WL "$VAR" > $OUT/p$i.out 2> $OUT/p$i.err
done
cd $REPO
python - "$OUT" "$KERNEL" "$TAG" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root, kern, tag = sys.argv[1:4]
agg = defaultdict(list)
for p in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== %s (%s) ==" % (tag, kern))
for c, v in sorted(agg.items()):
    print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
g = lambda k: (sum(agg[k]) / len(agg[k])) if agg.get(k) else float("nan")
print("   fabric read GB %.3f | mean read latency %.0f cyc | L2 hit (all lines) %.1f %% | wait_any/wave_cycles %.2f" % (
    g("TCC_EA0_RDREQ_sum") * 128 / 1e9, g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum"),
    100 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")))
PY
