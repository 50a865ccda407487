#!/bin/bash
# round 6, session 10 (EXPERIMENTS build): the window kernel with 64-channel slices (D3F_EXP_WINDOW_NARROW=1: 256-byte pool slots, one
# vector per lane, five waves per SIMD) against the 128-channel slices, same box; bit-identity tests with the knob set
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s10; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-46s step %.4f ms kernel avg %.4f min %.4f frac %.3f verified %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-500:])
PY
}
python -m d3fields_amd.build > $OUT/build_exp.log 2>&1
D3F_EXP_WINDOW_NARROW=1 timeout -k 5 900 python -m pytest tests/test_gpu_walks.py -x -q -m gpu -k "window or lattice_walk or cloud_gate or map_order" 2>&1 | tail -3 | cut -c1-200
for NAR in 0 1 0 1; do
  for SPEC in c2_patch c3_patch ref_patch c4_patch c2_patch:random ref_patch:random; do
    WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
    D3F_EXP_WINDOW_NARROW=$NAR timeout -k 5 300 python bench.py --no-cpu-baseline --traffic off --steps 30 --workload $WL --points $PTS > $OUT/n${NAR}_${WL}_$PTS.json 2> $OUT/n${NAR}_${WL}_$PTS.err
    line $OUT/n${NAR}_${WL}_$PTS.json "narrow=$NAR $WL $PTS"
  done
done
for KN in "D3F_EXP_WINDOW_WANT=15" "D3F_EXP_WINDOW_WANT=14" "D3F_EXP_WINDOW_OCC=4" "D3F_EXP_WINDOW_OCC=3"; do
  for WL in c4_patch c2_patch ref_patch; do
    env D3F_EXP_WINDOW_NARROW=1 $KN timeout -k 5 300 python bench.py --no-cpu-baseline --no-verify --traffic off --steps 30 --workload $WL > $OUT/k_${WL}.json 2> $OUT/k_${WL}.err
    line $OUT/k_${WL}.json "narrow=1 $KN $WL"
  done
done
