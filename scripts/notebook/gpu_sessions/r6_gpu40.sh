#!/bin/bash
# round 6, session 40: vector-L1 accesses of the distance-only kernel, lattice bricks vs caller order (one PMC pass each; EXPERIMENTS build for D3F_EXP_DIST=32)
set -u
REPO=$(pwd); TAG=${TAG:-r6_s40}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
python -m d3fields_amd.build > $OUT/build_exp.log 2>&1 || tail -20 $OUT/build_exp.log
cd /tmp
for D in ${DIST_LIST:-0 64 128 32}; do
  D3F_EXP_DIST=$D timeout -k 5 400 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE -d $OUT/d$D -o pmc --output-format csv -- python $REPO/bench.py --workload dist_only --steps 4 --warmup 1 --no-cpu-baseline --no-verify --traffic off > /dev/null 2> $OUT/d$D.err
  python - $OUT/d$D $D <<'PY'
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "fused_eval" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("D3F_EXP_DIST=%s" % sys.argv[2], {k: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for k, v in sorted(agg.items())})
PY
  rm -rf $OUT/d$D
done
