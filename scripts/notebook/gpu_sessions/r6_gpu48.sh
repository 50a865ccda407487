#!/bin/bash
# round 6, session 48 (PRODUCT build): the whole GPU suite on the sources with the distance-only kernel, then dist_only and the default bench line
set -u
REPO=$(pwd); TAG=${TAG:-r6_s48}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | cut -c1-250 | tee $OUT/suite.txt
for WL in dist_only c2_dense; do
  timeout -k 5 400 python bench.py --no-cpu-baseline --steps 30 --workload $WL > $OUT/$WL.json 2> $OUT/$WL.err
  python - $OUT/$WL.json $WL <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
print("%-12s step %.4f ms kernel avg %.4f frac %.3f traffic %s verified %s kernel %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r.get("traffic"), d.get("verified"), r.get("kernel")))
PY
done
