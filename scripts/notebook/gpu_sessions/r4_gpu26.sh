#!/bin/bash
# round 4, session 26 (EXPERIMENTS build): C4-patch (8 views, 117-slot pool at two workgroups per CU): how much does the pool size
# matter?  D3F_EXP_WINDOW_POOL = 80 ... 117; and phase stamps of the final kernel for C2-patch / C4-patch
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4y; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
B="python $REPO/bench.py --no-cpu-baseline --steps 30"
line() { python - $1 <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f frac %.3f verified %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print("ERR", e)
PY
}
for POOL in 0 110 100 90 80 0; do
    D3F_EXP_WINDOW_POOL=$POOL timeout -k 5 300 $B --workload c4_patch > $OUT/p_${POOL}.json 2> $OUT/p_${POOL}.err
    echo "c4_patch pool=$POOL: $(line $OUT/p_${POOL}.json)"
done
for POOL in 0 72 64 56; do
    D3F_EXP_WINDOW_POOL=$POOL timeout -k 5 300 $B --workload ref_patch > $OUT/r_${POOL}.json 2> $OUT/r_${POOL}.err
    echo "ref_patch pool=$POOL: $(line $OUT/r_${POOL}.json)"
done
D3F_EXP_STAMPS=1 timeout -k 5 300 python scripts/exp_stamps.py c2_patch c4_patch > $OUT/stamps.txt 2>&1; grep -v amdgpu $OUT/stamps.txt | head -60
