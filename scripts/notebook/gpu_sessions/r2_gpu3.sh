#!/bin/bash
# Round 2, GPU call 3: cell-run gather variants (vectors per lane x run length) on the patch-resolution workloads.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2c; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-verify --steps 20"
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -8 $OUT/pytest.log
run() { # tag workload env...
  TAG=$1; WL=$2; shift 2
  env "$@" timeout -k 5 200 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>&1
}
for WL in c2_patch c3_patch c4_patch; do
  run direct $WL D3F_EXP_RUNS=-1
  run auto $WL D3F_EXP_RUNS=0
  run u1k8occ5 $WL D3F_EXP_RUNS_U=1 D3F_EXP_RUNS_OCC=5
  run u3k2 $WL D3F_EXP_RUNS_U=3 D3F_EXP_RUNS=2
  run u3k4 $WL D3F_EXP_RUNS_U=3 D3F_EXP_RUNS=4
  run u2k4 $WL D3F_EXP_RUNS_U=2 D3F_EXP_RUNS=4
  run u2k8 $WL D3F_EXP_RUNS_U=2 D3F_EXP_RUNS=8
done
$B --workload c2_patch --points random > $OUT/bench_c2_patch_random_auto.json 2>&1
for f in $OUT/bench_*.json; do echo "$(basename $f): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    print("step %.3f ms kernel %.3f ms frac %.3f value %.3e %s | %s" % (d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["value"], d["roofline"]["kernel"], d["config"].get("point_order")))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
