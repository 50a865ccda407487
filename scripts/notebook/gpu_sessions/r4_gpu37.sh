#!/bin/bash
# round 4, session 37 (EXPERIMENTS build): the pipelined point loop with its corner reads TWO (three) steps ahead of their fma
# instead of one (D3F_EXP_WINDOW_PD=2 / 3; 157-161 VGPRs instead of 121): equality tests with the knob on, then bench lines
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ag; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
D3F_EXP_WINDOW_PD=2 timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest (pd=2) rc=$?"; tail -2 $OUT/pytest.log | cut -c1-160
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
for PD in 1 2 3; do
  for WL in c2_patch c3_patch c4_patch ref_patch; do
    D3F_EXP_WINDOW_PD=$PD timeout -k 5 300 $B --workload $WL > $OUT/p_${PD}_${WL}_$ROUND.json 2> $OUT/p_${PD}_${WL}_$ROUND.err
    echo "pd=$PD $WL: $(line $OUT/p_${PD}_${WL}_$ROUND.json)"
  done
done
done
