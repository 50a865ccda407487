#!/bin/bash
# round 4, session 18 (PRODUCT builds with -DD3F_WIN_ABLATE=bits, scripts/build_ablate.py): what-if timings of the window kernel
# (1 = no copies after slice 0, 2 = no point loop, 4 = no row stores; results are wrong by construction, only the kernel time is read)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4r; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-verify --steps 30"
for AB in 0 1 2 3 4 5 0; do
  cp $REPO/build_ab/ablate_$AB.so $REPO/d3fields_amd/libd3fields_hip.so
  for WL in c2_patch c4_patch ref_patch; do
    timeout -k 5 300 $B --workload $WL > $OUT/b_${AB}_${WL}.json 2> $OUT/b_${AB}_${WL}.err
    echo "ablate $AB $WL: $(python - $OUT/b_${AB}_${WL}.json <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"]))
except Exception as e:
    print("ERR", e)
PY
)"
  done
done
