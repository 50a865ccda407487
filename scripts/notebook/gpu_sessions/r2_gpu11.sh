#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2k; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "sliced" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>&1; }
for WL in c2_dense c3_dense; do
  run base $WL D3F_EXP_SLICED=0
  for SL in 1 2 3; do for VC in 2 4; do run sl${SL}vc${VC} $WL D3F_EXP_SLICED=$SL D3F_EXP_SLICED_VC=$VC; done; done
  run sl2vc1 $WL D3F_EXP_SLICED=2 D3F_EXP_SLICED_VC=1
done
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms (min %.3f) | frac %.3f | verified %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
