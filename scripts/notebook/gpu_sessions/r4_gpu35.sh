#!/bin/bash
# round 4, session 35 (EXPERIMENTS build): rows as `sc1 nt` stores (window kernel: always; the others: policy 2) against `nt` alone
# (D3F_EXP_STORE=3) on the dense / cloud / thin-map kernels; the walks + parity tests first
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4af; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
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
for ROUND in 1 2; do
for ST in 3 2; do
  for WL in c2_dense c3_dense c4_dense c3_patch c5_track; do
    D3F_EXP_STORE=$ST timeout -k 5 300 $B --workload $WL > $OUT/s_${ST}_${WL}_$ROUND.json 2> $OUT/s_${ST}_${WL}_$ROUND.err
    echo "store=$ST $WL: $(line $OUT/s_${ST}_${WL}_$ROUND.json)"
  done
  for WL in c2_dense c2_patch c4_patch; do
    D3F_EXP_STORE=$ST timeout -k 5 300 $B --workload $WL --points random > $OUT/s_${ST}_${WL}_cloud_$ROUND.json 2> $OUT/s_${ST}_${WL}_cloud_$ROUND.err
    echo "store=$ST $WL cloud: $(line $OUT/s_${ST}_${WL}_cloud_$ROUND.json)"
  done
done
done
