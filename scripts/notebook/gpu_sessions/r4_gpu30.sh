#!/bin/bash
# round 4, session 30 (EXPERIMENTS build): C4-patch (8 views, two workgroups per CU): 512 lanes per workgroup over the 64-point brick
# (D3F_EXP_WINDOW_NT=512: twice the waves per pool) against 256; equality tests with the knob on
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ac; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
D3F_EXP_WINDOW_NT=512 timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest (nt=512) rc=$?"; tail -2 $OUT/pytest.log | cut -c1-160
B="python $REPO/bench.py --no-cpu-baseline --steps 30"
line() { python - $1 <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f frac %.3f verified %s %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"][-32:]))
except Exception as e:
    print("ERR", e)
PY
}
for ROUND in 1 2 3; do
for NT in 256 512; do
    D3F_EXP_WINDOW_NT=$NT timeout -k 5 300 $B --workload c4_patch > $OUT/n_${NT}_$ROUND.json 2> $OUT/n_${NT}_$ROUND.err
    echo "c4_patch lanes=$NT: $(line $OUT/n_${NT}_$ROUND.json)"
done
done
