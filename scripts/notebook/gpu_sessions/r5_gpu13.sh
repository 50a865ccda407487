#!/bin/bash
# round 5, session 13 (EXPERIMENTS what-if builds -DD3F_SLICED_WHATIF=bits): how much of the C2-dense kernel is phase A, how much the gather?
#   0 the kernel; 1 phase A without depth lookup / validity / weight; 4 no gather (phase A + stores); 5 both
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r5_s13; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
for AB in 0 1 4 5 0; do
  cp $REPO/build_ab/sliced_$AB.so $REPO/d3fields_amd/libd3fields_hip.so
  for WL in c2_dense c3_dense; do
    timeout -k 5 300 python bench.py --no-cpu-baseline --no-verify --steps 30 --workload $WL > $OUT/b_${AB}_$WL.json 2> $OUT/b_${AB}_$WL.err
    python - $OUT/b_${AB}_$WL.json $AB $WL <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("whatif %s %-9s kernel avg %.4f min %.4f %s" % (sys.argv[2], sys.argv[3], r["kernel_ms_avg"], r["kernel_ms_min"], r["kernel"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  done
done
