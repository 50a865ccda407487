#!/bin/bash
# round 6, sessions 16+ (EXPERIMENTS build): the register-rows kernel, iteration -- phase stamps, the 1024-channel bench lines
# (ROWS_LIST="0 -1" adds the other kernels on the same box), optionally the tests (TESTS=1)
set -u
REPO=$(pwd); TAG=${TAG:-r6_s16}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-46s step %.4f ms kernel avg %.4f min %.4f frac %.3f verified %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
python -m d3fields_amd.build > $OUT/build_exp.log 2>&1 || tail -20 $OUT/build_exp.log
[ "${TESTS:-0}" = 1 ] && timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -q -m gpu -k "c4 or ref_patch or 1024 or bench_workload" 2>&1 | tail -8 | cut -c1-220
D3F_EXP_STAMPS=1 python scripts/notebook/exp_stamps.py ${STAMP_WL:-c4_patch ref_patch} 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps.txt
for ROWS in ${ROWS_LIST:-0}; do
  for SPEC in ${SPECS:-c4_patch ref_patch c4_patch:random ref_patch:random ref_patch:surface}; do
    WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
    D3F_EXP_ROWS=$ROWS timeout -k 5 300 python bench.py --no-cpu-baseline --traffic off --steps 30 --workload $WL --points $PTS > $OUT/r${ROWS}_${WL}_$PTS.json 2> $OUT/r${ROWS}_${WL}_$PTS.err
    line $OUT/r${ROWS}_${WL}_$PTS.json "rows=$ROWS $WL $PTS"
  done
done
