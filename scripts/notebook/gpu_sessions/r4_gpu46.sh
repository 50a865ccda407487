#!/bin/bash
# round 4, session 46 (PRODUCT builds): the thin maps' gather of the window kernel in FRONT of the slices (underneath slice 0's copies)
# instead of behind them: the walks / parity tests, then same-box A/B on the workloads with thin maps
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4al; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log | cut -c1-160
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
for ROUND in 1 2 3; do
for LIB in thin_last thin_first; do
  cp $REPO/build_ab/$LIB.so $REPO/d3fields_amd/libd3fields_hip.so
  for WL in c3_patch ref_patch; do
    timeout -k 5 300 $B --workload $WL > $OUT/${LIB}_${WL}_$ROUND.json 2> $OUT/${LIB}_${WL}_$ROUND.err
    echo "$LIB $WL: $(line $OUT/${LIB}_${WL}_$ROUND.json)"
  done
done
done
