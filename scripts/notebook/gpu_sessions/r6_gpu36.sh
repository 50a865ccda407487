#!/bin/bash
# round 6, session 36+ (PRODUCT build): the distance-only kernel, iteration -- its tests, then bench lines of dist_only (x REPS)
set -u
REPO=$(pwd); TAG=${TAG:-r6_s36}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-30s step %.4f ms kernel avg %.4f min %.4f frac %.3f verified %s kernel %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r.get("kernel")))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
[ "${TESTS:-1}" = 1 ] && timeout -k 5 600 python -m pytest tests/test_gpu_dist.py -q -m gpu 2>&1 | tail -5 | cut -c1-250
for R in $(seq 1 ${REPS:-2}); do
  for WL in ${WLS:-dist_only}; do
    timeout -k 5 300 python bench.py --no-cpu-baseline --traffic off --steps 30 --workload $WL > $OUT/${WL}_$R.json 2> $OUT/${WL}_$R.err
    line $OUT/${WL}_$R.json "$WL #$R"
  done
done
