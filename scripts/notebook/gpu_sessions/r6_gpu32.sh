#!/bin/bash
# round 6, sessions 32 / 34 (EXPERIMENTS build): the distance-only pass as its own kernel (fused_eval_dist_kernel: KRt in SGPRs, the views of a
# point in flight together, 8 or 6 waves per SIMD) against the branch of fused_eval_kernel (D3F_EXP_DIST=-1), same box; + 16: the compiler's divisions instead of the short form; parity tests first
set -u
REPO=$(pwd); TAG=${TAG:-r6_s32}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-46s step %.4f ms kernel avg %.4f min %.4f frac %.3f verified %s kernel %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r.get("kernel")))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
python -m d3fields_amd.build > $OUT/build_exp.log 2>&1 || tail -20 $OUT/build_exp.log
for D in ${DIST_LIST:-0 6}; do
  D3F_EXP_DIST=$D timeout -k 5 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "${TESTK:-golden or reference or seeded or directed or dist}" 2>&1 | tail -4 | cut -c1-220
done
for D in ${BENCH_LIST:--1 0 6 -1 0 6}; do
  for WL in ${WLS:-dist_only}; do
    D3F_EXP_DIST=$D timeout -k 5 300 python bench.py --no-cpu-baseline --traffic off --steps 30 --workload $WL > $OUT/d${D}_$WL.json 2> $OUT/d${D}_$WL.err
    line $OUT/d${D}_$WL.json "dist=$D $WL"
  done
done
