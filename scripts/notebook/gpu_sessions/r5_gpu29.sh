#!/bin/bash
# round 5, session 29 (EXPERIMENTS what-if builds: scripts/notebook/patches/r5_s29_window_whatif_skip.patch applied, -DD3F_WIN_ABLATE=32/64/96): the window kernel with every 3rd / 2nd / 4th (point, view)
# step reusing the previous step's corner vectors instead of reading its own from the pool -- the best a wave-uniform
# "same texel cells as the point before" skip could reach (results wrong by construction; only times are read)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r5_s29; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
for AB in 0 96 32 64 0; do
  cp $REPO/build_ab/ablate_$AB.so $REPO/d3fields_amd/libd3fields_hip.so
  for WL in c2_patch c4_patch ref_patch; do
    timeout -k 5 300 python bench.py --no-cpu-baseline --no-verify --steps 30 --workload $WL > $OUT/b_${AB}_$WL.json 2> $OUT/b_${AB}_$WL.err
    python - $OUT/b_${AB}_$WL.json $AB $WL <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("whatif %s %-9s kernel avg %.4f min %.4f %s" % (sys.argv[2], sys.argv[3], r["kernel_ms_avg"], r["kernel_ms_min"], r["kernel"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  done
done
