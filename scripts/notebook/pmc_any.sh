#!/bin/bash
# scripts/pmc_any.sh <tag> <kernel-name substring> <command...> : SQ counter passes (one small set per pass) of one kernel
set -u
TAG=$1; KERNEL=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o pmc --output-format csv -- "$@" > /dev/null 2> $OUT/p$i.err
done
cd $REPO
python - "$OUT" "$KERNEL" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root, kern = sys.argv[1], sys.argv[2]
agg = defaultdict(list)
for p in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print("   %-26s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
