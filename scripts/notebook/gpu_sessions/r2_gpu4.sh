#!/bin/bash
# Round 2, GPU call 4: occupancy variants of the cell-run gather, finer Morton cells for clouds, c4_dense evidence.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2d; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-verify --steps 20"
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "cell_run" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>&1; }
for WL in c2_patch c3_patch c4_patch; do
  run occ5 $WL D3F_EXP_RUNS_OCC=5
  run occ6 $WL D3F_EXP_RUNS_OCC=6
  run occ4 $WL D3F_EXP_RUNS_OCC=4
  run k4occ6 $WL D3F_EXP_RUNS=4
done
env timeout -k 5 300 $B --workload c4_patch --tuning 0x1000000 > $OUT/bench_c4_patch_fine1.json 2>&1
env timeout -k 5 300 $B --workload c4_patch --tuning 0x2000000 > $OUT/bench_c4_patch_fine2.json 2>&1
env timeout -k 5 300 $B --workload c2_patch --points random --tuning 0x2000000 > $OUT/bench_c2_patch_random_fine2.json 2>&1
env timeout -k 5 300 $B --workload c2_patch --points random > $OUT/bench_c2_patch_random.json 2>&1
env timeout -k 5 400 $B --workload c4_dense > $OUT/bench_c4_dense.json 2> $OUT/bench_c4_dense.err
for f in $OUT/bench_*.json; do echo "$(basename $f): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    print("step %.3f ms kernel %.3f ms frac %.3f value %.3e %s | %s" % (d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["value"], d["roofline"]["kernel"], d["config"].get("point_order")))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
