#!/bin/bash
# round 4, session 25 (PRODUCT build): non-temporal rows.  GPU suite, every bench line, the default line with the CPU baseline,
# then rocprofv3 summaries + counter passes of the workloads whose kernels changed
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_v3; mkdir -p $OUT/bench
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-200
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
for WL in c2_dense c3_dense c4_dense c2_patch c3_patch c4_patch ref_patch dist_only c5_track c2_dense_f16 c2_patch_f16; do
    timeout -k 5 300 $B --workload $WL > $OUT/bench/${WL}.json 2> $OUT/bench/${WL}.err
    echo "$WL: $(line $OUT/bench/${WL}.json)"
done
for WL in c2_dense c4_patch; do
    timeout -k 5 300 $B --workload $WL --points random > $OUT/bench/${WL}_cloud.json 2> $OUT/bench/${WL}_cloud.err
    echo "$WL cloud: $(line $OUT/bench/${WL}_cloud.json)"
done
timeout -k 5 300 $B --workload c2_dense --refresh-maps > $OUT/bench/c2_dense_refresh.json 2> $OUT/bench/c2_dense_refresh.err; echo "c2_dense refresh: $(line $OUT/bench/c2_dense_refresh.json)"
timeout -k 5 600 python $REPO/bench.py > $OUT/default_bench.json 2> $OUT/default_bench.err; echo "default: $(line $OUT/default_bench.json)"
bash $REPO/scripts/r4_profile_all.sh r4_v3 c2_dense c3_dense c2_patch c3_patch c4_patch ref_patch > $OUT/profile_all.log 2>&1
rm -rf $OUT/*/trace $OUT/*/pmc_* 2>/dev/null
ls $OUT
