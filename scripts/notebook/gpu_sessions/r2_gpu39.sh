#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3j; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL --points random > $OUT/bench_${WL}_$TAG.json 2>$OUT/bench_${WL}_$TAG.err; }
for WL in c2_patch c3_patch; do
  run runs $WL D3F_EXP_WINDOW=0
  run w64 $WL D3F_EXP_WINDOW=64
  run w32 $WL D3F_EXP_WINDOW=32
  run w64o3 $WL D3F_EXP_WINDOW=64 D3F_EXP_WINDOW_OCC=3
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
