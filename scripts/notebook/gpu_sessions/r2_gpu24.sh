#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2x; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>$OUT/bench_${WL}_$TAG.err; }
run base c4_patch D3F_EXP_RUNS_OCC=0
run occ3 c4_patch D3F_EXP_RUNS_OCC=3
run base2 c4_patch D3F_EXP_RUNS_OCC=0
run occ3b c4_patch D3F_EXP_RUNS_OCC=3
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
