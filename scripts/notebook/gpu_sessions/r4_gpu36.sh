#!/bin/bash
# round 4, session 36 (PRODUCT build): window kernel rows as `sc1 nt`, every other kernel's rows `nt`: GPU suite, the window
# workloads' bench lines, their rocprofv3 summaries + counters again (-> profiles/r4_v3)
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
for WL in c2_patch c3_patch c4_patch ref_patch c2_dense c3_dense; do
    timeout -k 5 300 $B --workload $WL > $OUT/bench/${WL}.json 2> $OUT/bench/${WL}.err
    echo "$WL: $(line $OUT/bench/${WL}.json)"
done
bash $REPO/scripts/r4_profile_all.sh r4_v3 c2_patch c3_patch c4_patch ref_patch > $OUT/profile_all3.log 2>&1
rm -rf $OUT/*/trace $OUT/*/pmc_* 2>/dev/null
