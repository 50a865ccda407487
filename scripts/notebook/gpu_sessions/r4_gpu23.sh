#!/bin/bash
# round 4, session 23 (EXPERIMENTS builds): non-temporal row stores.  (a) window kernel, ONE launch (no split): build with
# -DD3F_WIN_ABLATE=32 (nt stores in the pipelined loop; results are right) against the plain build; (b) the dense kernels:
# D3F_EXP_STORE=2 (nt) against the default (sc1)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4w; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
B="python $REPO/bench.py --no-cpu-baseline --steps 30"
line() { python - $1 <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f frac %.3f verified %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print("ERR", e)
PY
}
for ROUND in 1 2; do
for AB in 0 32; do
  cp $REPO/build_ab/ablate_$AB.so $REPO/d3fields_amd/libd3fields_hip.so
  for WL in c2_patch c3_patch c4_patch ref_patch; do
    timeout -k 5 300 $B --workload $WL > $OUT/b_${AB}_${WL}_$ROUND.json 2> $OUT/b_${AB}_${WL}_$ROUND.err
    echo "window nt=$AB $WL: $(line $OUT/b_${AB}_${WL}_$ROUND.json)"
  done
done
done
cp $REPO/build_ab/ablate_0.so $REPO/d3fields_amd/libd3fields_hip.so
for ROUND in 1 2; do
for ST in 0 2; do
  for WL in c2_dense c3_dense c4_dense; do
    D3F_EXP_STORE=$ST timeout -k 5 300 $B --workload $WL > $OUT/d_${ST}_${WL}_$ROUND.json 2> $OUT/d_${ST}_${WL}_$ROUND.err
    echo "dense store=$ST $WL: $(line $OUT/d_${ST}_${WL}_$ROUND.json)"
  done
  D3F_EXP_STORE=$ST timeout -k 5 300 $B --workload c2_dense --points cloud > $OUT/d_${ST}_c2cloud_$ROUND.json 2> $OUT/d_${ST}_c2cloud_$ROUND.err
  echo "dense store=$ST c2_dense cloud: $(line $OUT/d_${ST}_c2cloud_$ROUND.json)"
  D3F_EXP_STORE=$ST timeout -k 5 300 $B --workload c4_patch --points cloud > $OUT/d_${ST}_c4pcloud_$ROUND.json 2> $OUT/d_${ST}_c4pcloud_$ROUND.err
  echo "runs store=$ST c4_patch cloud: $(line $OUT/d_${ST}_c4pcloud_$ROUND.json)"
done
done
