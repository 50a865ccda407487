#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2t; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python scripts/exp_window.py c2_patch > $OUT/window.log 2>&1; echo "rc=$?" >> $OUT/window.log
cat $OUT/window.log | tail -22
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c2_dense c2_patch c3_dense c3_patch c4_patch c5_track; do
  timeout -k 5 400 $B --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
done
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms | %.3e pts/s | frac %.3f | verified %s | %s" % (d["ms_per_step"], r["kernel_ms_avg"], d["value"], r["frac"], d.get("verified"), r["kernel"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
