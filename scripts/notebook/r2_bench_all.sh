#!/bin/bash
# One bench.py line per single-GPU workload (what DESIGN.md section 6 tabulates): gpurun_out/<tag>/bench_<workload>.json
set -u
TAG=${1:-r2_bench}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c2_dense c2_patch c3_dense c3_patch c4_patch c4_dense c5_track c2_dense_f16 c2_patch_f16; do
  timeout -k 5 400 $B --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
done
for WL in c2_dense c2_patch c3_patch; do
  timeout -k 5 400 $B --workload $WL --points random > $OUT/bench_${WL}_random.json 2> $OUT/bench_${WL}_random.err
done
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms | %.3e pts/s | cached-order %.3e | frac %.3f | verified %s | %s | %s" % (d["ms_per_step"], r["kernel_ms_avg"], d["value"], d.get("points_per_s_with_cached_point_order", 0), r["frac"], d.get("verified"), r["kernel"], d["config"].get("point_order")))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
