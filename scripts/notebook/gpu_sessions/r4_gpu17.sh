#!/bin/bash
# round 4, session 17 (PRODUCT build): same-box A/B of the window kernel's set-up (build_ab/head.so = the last commit, build_ab/new.so =
# corner loads early + window boxes per wave + projection/depth lookup ahead of the slot table), then the whole GPU suite on the new one
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4q; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 30"
for ROUND in 1 2; do
  for WHICH in head new; do
    cp $REPO/build_ab/$WHICH.so $REPO/d3fields_amd/libd3fields_hip.so
    for WL in c2_patch c3_patch c4_patch ref_patch; do
        timeout -k 5 300 $B --workload $WL > $OUT/b_${WHICH}_${WL}_$ROUND.json 2> $OUT/b_${WHICH}_${WL}_$ROUND.err
        echo "$WHICH $ROUND $WL: $(python - $OUT/b_${WHICH}_${WL}_$ROUND.json <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f frac %.3f verified %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print("ERR", e)
PY
)"
    done
  done
done
cp $REPO/build_ab/new.so $REPO/d3fields_amd/libd3fields_hip.so
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-200
