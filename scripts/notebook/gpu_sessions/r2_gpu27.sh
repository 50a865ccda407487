#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2z; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>$OUT/bench_${WL}_$TAG.err; }
for T in 0 32 128; do run t$T c5_track D3F_EXP_RUNS_TILE=$T; run t$T c2_patch D3F_EXP_RUNS_TILE=$T; done
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms (min %.3f) | %.3e pts/s | verified %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], d["value"], d.get("verified")))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
