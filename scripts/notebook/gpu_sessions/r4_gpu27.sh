#!/bin/bash
# round 4, session 27 (EXPERIMENTS build): with non-temporal rows, 4 workgroups per CU (54-slot pools: D3F_EXP_WINDOW_WANT=13 slots per view) against 3 (80 slots) again
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4z; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
B="python $REPO/bench.py --no-cpu-baseline --steps 30"
line() { python - $1 <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f frac %.3f verified %s %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"][-40:]))
except Exception as e:
    print("ERR", e)
PY
}
for ROUND in 1 2; do
for OCC in 0 13; do
  for WL in c2_patch c3_patch ref_patch; do
    D3F_EXP_WINDOW_WANT=$OCC timeout -k 5 300 $B --workload $WL > $OUT/o_${OCC}_${WL}_$ROUND.json 2> $OUT/o_${OCC}_${WL}_$ROUND.err
    echo "slots per view wanted=$OCC $WL: $(line $OUT/o_${OCC}_${WL}_$ROUND.json)"
  done
done
done
