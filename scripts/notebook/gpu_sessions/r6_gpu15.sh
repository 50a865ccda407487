#!/bin/bash
# round 6, session 15 (EXPERIMENTS build): the register-rows kernel with counted waits on asm loads and ops read one ahead --
# oracle / bit-identity tests of the 1024-channel workloads, then the bench lines (session 14 has the other kernels on these workloads)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s15; mkdir -p $OUT
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
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -q -m gpu -k "c4 or ref_patch or 1024 or bench_workload" 2>&1 | tail -12 | cut -c1-220
for ROWS in ${ROWS_LIST:-0 0}; do
  for SPEC in c4_patch ref_patch c4_patch:random ref_patch:random ref_patch:surface; do
    WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
    D3F_EXP_ROWS=$ROWS timeout -k 5 300 python bench.py --no-cpu-baseline --traffic off --steps 30 --workload $WL --points $PTS > $OUT/r${ROWS}_${WL}_$PTS.json 2> $OUT/r${ROWS}_${WL}_$PTS.err
    line $OUT/r${ROWS}_${WL}_$PTS.json "rows=$ROWS $WL $PTS"
  done
done
