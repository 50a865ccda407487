#!/bin/bash
# scripts/pmc_kernel.sh <tag> <bench args...> : SQ / LDS counter passes of the fused kernel
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline $*"
cd /tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/p$i.err
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for p in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "fused_eval" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-26s %16.0f" % (c, sum(v) / len(v)))
PY
