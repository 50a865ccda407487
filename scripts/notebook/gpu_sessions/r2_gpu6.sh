#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2f; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-verify --steps 20"
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py tests/test_gpu_callers.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>&1; }
for WL in c2_dense c3_dense c2_patch; do
  run base $WL D3F_EXP_STORE=0
  run sc1 $WL D3F_EXP_STORE=1
done
run t124 c2_dense D3F_EXP_WALK_TILE=124
run t142 c2_dense D3F_EXP_WALK_TILE=142
run t214 c2_dense D3F_EXP_WALK_TILE=214
run t241 c2_dense D3F_EXP_WALK_TILE=241
run t422 c2_dense D3F_EXP_WALK_TILE=421
run t118 c2_dense D3F_EXP_WALK_TILE=118
timeout -k 5 300 $B --workload c2_dense --tuning 0x20000000 > $OUT/bench_c2_dense_chunk1024.json 2>&1
timeout -k 5 300 $B --workload c2_dense --tuning 0x40000000 > $OUT/bench_c2_dense_chunk2048.json 2>&1
timeout -k 5 300 $B --workload c2_dense --tuning 0x80000000 > $OUT/bench_c2_dense_chunk8192.json 2>&1
timeout -k 5 300 $B --workload c2_dense --tuning 0xa0000000 > $OUT/bench_c2_dense_chunk16384.json 2>&1
for f in $OUT/bench_*.json; do echo "$(basename $f): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    print("step %.3f ms kernel %.3f ms (min %.3f) frac %.3f value %.3e %s" % (d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["kernel_ms_min"], d["roofline"]["frac"], d["value"], d["roofline"]["kernel"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
