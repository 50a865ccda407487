#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3a; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>$OUT/bench_${WL}_$TAG.err; }
for WL in c2_patch c3_patch; do
  run base $WL D3F_EXP_WINDOW=0
  run w64o4 $WL D3F_EXP_WINDOW=64 D3F_EXP_WINDOW_OCC=4
  run w64o5 $WL D3F_EXP_WINDOW=64 D3F_EXP_WINDOW_OCC=5
  run w32o5 $WL D3F_EXP_WINDOW=32 D3F_EXP_WINDOW_OCC=5
  run w32o6 $WL D3F_EXP_WINDOW=32 D3F_EXP_WINDOW_OCC=6
  run w32o7 $WL D3F_EXP_WINDOW=32 D3F_EXP_WINDOW_OCC=7
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
