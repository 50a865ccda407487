#!/bin/bash
# scripts/pmc_slice.sh : L2 hit/miss + fabric bytes of channel-sliced launches
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_slice; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for CFG in "96 0x500" "48 0x600"; do
  set -- $CFG
  i=0
  for PMC in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do   # one small counter set per pass (hardware limit)
    i=$((i+1))
    timeout -k 5 70 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/c$1/p$i -o pmc --output-format csv -- python $REPO/scripts/exp_slice_pmc.py $1 $2 > /dev/null 2> $OUT/c$1_p$i.err
  done
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "c*"))):
    if not os.path.isdir(d): continue
    agg = defaultdict(list)
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "fused_eval" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(agg.items())})
PY
