#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2h; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-verify --steps 20"
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_callers.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>&1; }
for WL in c2_patch c3_patch c4_patch; do
  run direct $WL D3F_EXP_RUNS=-1
  run k4occ6 $WL D3F_EXP_RUNS=4
  run k4occ7 $WL D3F_EXP_RUNS=4 D3F_EXP_RUNS_OCC=7
  run k8occ5 $WL D3F_EXP_RUNS=8 D3F_EXP_RUNS_OCC=5
  run k8occ6 $WL D3F_EXP_RUNS=8 D3F_EXP_RUNS_OCC=6
  run u2k4 $WL D3F_EXP_RUNS_U=2 D3F_EXP_RUNS=4
  run u3k4 $WL D3F_EXP_RUNS_U=3 D3F_EXP_RUNS=4
  run u3k2 $WL D3F_EXP_RUNS_U=3 D3F_EXP_RUNS=2
  run u2k8 $WL D3F_EXP_RUNS_U=2 D3F_EXP_RUNS=8
done
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms (min %.3f) | frac %.3f | %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], r["kernel"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
