#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2l; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 1 --no-verify"
cd /tmp
for VAR in "0 1" "2 1" "3 4" "3 2"; do
  set -- $VAR
  for PMC in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
    N=$(echo $PMC | tr ' ' '_' | cut -c1-40)
    D3F_EXP_SLICED=$1 D3F_EXP_SLICED_VC=$2 timeout -k 5 120 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc_sl$1vc$2/$N -o pmc --output-format csv -- $B --workload c2_dense > /dev/null 2> $OUT/pmc_sl$1vc$2_$N.err
  done
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for var in sorted(glob.glob(os.path.join(root, "pmc_sl*"))):
    if not os.path.isdir(var): continue
    agg = defaultdict(lambda: defaultdict(list))
    for p in glob.glob(os.path.join(var, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "fused_eval" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(os.path.basename(var), k)
        for c, v in sorted(cs.items()):
            print("   %-34s %18.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
