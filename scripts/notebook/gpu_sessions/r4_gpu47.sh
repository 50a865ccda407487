#!/bin/bash
# round 4, session 47 (PRODUCT builds): what-if -DD3F_THIN_WHATIF -- the thin maps' gather with every lane reading its view's first
# texel (same instructions, one cache line per load instruction): is that gather bound by cache-line look-ups?  (results wrong)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4am; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-verify --steps 30"
line() { python - $1 <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"]))
except Exception as e:
    print("ERR", e)
PY
}
for ROUND in 1 2; do
for LIB in plain thin_one_texel; do
  cp $REPO/build_ab/$LIB.so $REPO/d3fields_amd/libd3fields_hip.so
  for WL in c3_patch ref_patch c3_dense; do
    timeout -k 5 300 $B --workload $WL > $OUT/${LIB}_${WL}_$ROUND.json 2> $OUT/${LIB}_${WL}_$ROUND.err
    echo "$LIB $WL: $(line $OUT/${LIB}_${WL}_$ROUND.json)"
  done
done
done
