#!/bin/bash
# round 4, session 34 (PRODUCT builds with -DD3F_ROW_STORE_BITS="..."): the window kernel's row stores with other cache-policy bits
# than `nt` alone (sc1 nt / sc0 sc1 nt / sc0 nt / sc0 sc1); results stay right, only the pipelined loop's stores change
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ae; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 30"
line() { python - $1 <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("kernel %.3f min %.3f frac %.3f verified %s" % (r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print("ERR", e)
PY
}
for LIB in nt sc1_nt sc0_sc1_nt sc0_nt sc0_sc1 nt; do
  cp $REPO/build_ab/$LIB.so $REPO/d3fields_amd/libd3fields_hip.so
  for WL in c2_patch c4_patch ref_patch; do
    timeout -k 5 300 $B --workload $WL > $OUT/${LIB}_${WL}.json 2> $OUT/${LIB}_${WL}.err
    echo "$LIB $WL: $(line $OUT/${LIB}_${WL}.json)"
  done
done
