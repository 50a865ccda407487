#!/bin/bash
# round 3, session 46: fabric read requests of C2-dense at texel strides of 1536 and 2048 bytes (does the bigger effective L2 remove fills?)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_l2probe; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for ST in 384 512; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_stride_$ST -o pmc --output-format csv -- python $REPO/scripts/exp_texel_stride.py c2_dense $ST > $OUT/stride_$ST.txt 2> $OUT/stride_$ST.err
done
cd $REPO
python - <<'PY'
import csv, glob
for st in (384, 512):
    agg = {}
    for p in glob.glob("gpurun_out/r3_l2probe/pmc_stride_%d/**/*counter_collection.csv" % st, recursive=True):
        for r in csv.DictReader(open(p)):
            if "sliced" in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    g = {k: sum(v) / len(v) for k, v in agg.items()}
    if g:
        print("texel stride %d floats: fabric reads %.2f GB per launch, L2 hit (all lines) %.1f %%  (n=%d)" % (
            st, g["TCC_EA0_RDREQ_sum"] * 128 / 1e9, 100 * g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"]), len(agg["TCC_HIT_sum"])))
PY
