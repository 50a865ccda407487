#!/bin/bash
# round 6, session 42 (EXPERIMENTS build): the distance-only kernel on the lattice -- lane orders / brick shapes, bench lines (D3F_EXP_DIST bits)
set -u
REPO=$(pwd); TAG=${TAG:-r6_s42}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-30s step %.4f ms kernel avg %.4f min %.4f frac %.3f verified %s kernel %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r.get("kernel")))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
python -m d3fields_amd.build > $OUT/build_exp.log 2>&1 || tail -20 $OUT/build_exp.log
for D in ${DIST_LIST:-32 0 64 128 256 32 0 64 128 256}; do
  D3F_EXP_DIST=$D timeout -k 5 300 python bench.py --no-cpu-baseline --traffic off --steps 30 --workload dist_only > $OUT/d$D.json 2> $OUT/d$D.err
  line $OUT/d$D.json "dist=$D"
done
