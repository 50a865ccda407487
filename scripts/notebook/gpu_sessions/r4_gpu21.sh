#!/bin/bash
# round 4, session 21 (EXPERIMENTS build): 256-byte slices in two pool buffers (D3F_EXP_WINDOW_DB=1), alone and with the split launch
# (D3F_EXP_WINDOW_SPLIT=1): equality tests, then bench lines of the four combinations
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4u; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
D3F_EXP_WINDOW_DB=1 timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_db.log 2>&1; echo "pytest (db) rc=$?"; tail -3 $OUT/pytest_db.log | cut -c1-200
D3F_EXP_WINDOW_DB=1 D3F_EXP_WINDOW_SPLIT=1 timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_db_split.log 2>&1; echo "pytest (db+split) rc=$?"; tail -3 $OUT/pytest_db_split.log | cut -c1-200
B="python $REPO/bench.py --no-cpu-baseline --steps 30"
for ROUND in 1 2; do
for MODE in 00 10 01 11; do
  for WL in c2_patch c3_patch c4_patch ref_patch; do
    D3F_EXP_WINDOW_DB=${MODE:0:1} D3F_EXP_WINDOW_SPLIT=${MODE:1:1} timeout -k 5 300 $B --workload $WL > $OUT/b_${MODE}_${WL}_$ROUND.json 2> $OUT/b_${MODE}_${WL}_$ROUND.err
    echo "db/split $MODE $WL: $(python - $OUT/b_${MODE}_${WL}_$ROUND.json <<'PY'
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
