#!/bin/bash
# round 4, session 24 (EXPERIMENTS build): non-temporal rows everywhere (window kernel: always; the other kernels: store policy 2 by
# default).  GPU suite, then bench lines: all workloads; the window kernel's split launch / two pool buffers again on top of it
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4x; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-200
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
for WL in c2_dense c3_dense c4_dense c2_patch c3_patch c4_patch ref_patch dist_only c5_track c2_dense_f16 c2_patch_f16; do
    timeout -k 5 300 $B --workload $WL > $OUT/b_${WL}.json 2> $OUT/b_${WL}.err
    echo "$WL: $(line $OUT/b_${WL}.json)"
done
for WL in c2_dense c4_patch c2_patch; do
  for ST in 2 1; do
    D3F_EXP_STORE=$ST timeout -k 5 300 $B --workload $WL --points random > $OUT/c_${ST}_${WL}.json 2> $OUT/c_${ST}_${WL}.err
    echo "cloud store=$ST $WL: $(line $OUT/c_${ST}_${WL}.json)"
  done
done
for MODE in 01 10 11 00; do
  for WL in c2_patch c3_patch c4_patch ref_patch; do
    D3F_EXP_WINDOW_DB=${MODE:0:1} D3F_EXP_WINDOW_SPLIT=${MODE:1:1} timeout -k 5 300 $B --workload $WL > $OUT/m_${MODE}_${WL}.json 2> $OUT/m_${MODE}_${WL}.err
    echo "db/split $MODE $WL: $(line $OUT/m_${MODE}_${WL}.json)"
  done
done
