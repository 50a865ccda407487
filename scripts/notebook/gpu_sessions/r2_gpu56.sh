#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3y; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "record_scratch or sliced or bench_workload or map_order" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -8 $OUT/pytest.log
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>$OUT/bench_${WL}_$TAG.err; }
for i in 1 2; do
  run one$i c2_dense D3F_EXP_SLICED_RECORDS=-1
  run two$i c2_dense D3F_EXP_SLICED_RECORDS=0
  run one$i c3_dense D3F_EXP_SLICED_RECORDS=-1
  run two$i c3_dense D3F_EXP_SLICED_RECORDS=0
done
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms (min %.3f) | frac %.3f | verified %s | %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
