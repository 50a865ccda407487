#!/bin/bash
# round 6, session 8: (PRODUCT) the new bench-line test + GPU suite; (EXPERIMENTS) C3-dense against the sliced kernel's geometry knobs
# (tile points 16 / 32 / 64 = brick 2x2x{1,2,4} x 4 tiles, unit size, interleave) -- same box
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s8; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_sharding.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-44s step %.4f ms kernel avg %.4f min %.4f frac %.3f  tile %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d["config"].get("tile_points")))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-400:])
PY
}
export D3F_BUILD_EXPERIMENTS=1
python -m d3fields_amd.build > $OUT/build_exp.log 2>&1
for WL in c3_dense c2_dense; do
  for KN in "" "D3F_EXP_SLICED_TILE=16" "D3F_EXP_SLICED_TILE=32" "D3F_EXP_SLICED_TILE=64" "D3F_EXP_SLICED_UNIT=64" "D3F_EXP_SLICED_UNIT=256" "D3F_EXP_SLICED_ILV=2" "D3F_EXP_SLICED_VC=4" "D3F_EXP_SLICED_VC=1" ""; do
    env $KN timeout -k 5 300 python bench.py --no-cpu-baseline --no-verify --traffic off --steps 30 --workload $WL > $OUT/${WL}_x.json 2> $OUT/${WL}_x.err
    line $OUT/${WL}_x.json "$WL ${KN:-default}"
  done
done
