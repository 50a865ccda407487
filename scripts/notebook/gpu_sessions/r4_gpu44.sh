#!/bin/bash
# round 4, session 44 (PRODUCT builds): what-if -DD3F_NT_POINT_OUTPUTS -- 'dist' / 'valid_mask' stored non-temporally (5 B per point:
# 616 MB in the distance-only pass over 123.2 M points)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ak; mkdir -p $OUT
export TMPDIR=/tmp
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
for ROUND in 1 2; do
for LIB in plain nt_points; do
  cp $REPO/build_ab/$LIB.so $REPO/d3fields_amd/libd3fields_hip.so
  for WL in dist_only c2_dense c3_dense; do
    timeout -k 5 300 $B --workload $WL > $OUT/${LIB}_${WL}_$ROUND.json 2> $OUT/${LIB}_${WL}_$ROUND.err
    echo "$LIB $WL: $(line $OUT/${LIB}_${WL}_$ROUND.json)"
  done
done
done
