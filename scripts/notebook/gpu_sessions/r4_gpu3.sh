#!/bin/bash
# round 4, session 3 (EXPERIMENTS build): software-pipelined point loop of the window kernel (records + corners of the next
# steps in flight behind the arithmetic), padded record stride: on / off, with / without deferred rows; tests; counters
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4c; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log | cut -c1-200
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c2_patch c3_patch c4_patch; do
  for PIPE in 0 -1; do for DEF in 0 1; do
    D3F_EXP_WINDOW_PIPE=$PIPE D3F_EXP_WINDOW_DEFER=$DEF timeout -k 5 300 $B --workload $WL > $OUT/b_${WL}_pipe${PIPE}_def${DEF}.json 2> $OUT/b_${WL}_pipe${PIPE}_def${DEF}.err
    echo "$WL pipe=$PIPE deferknob=$DEF: $(python - $OUT/b_${WL}_pipe${PIPE}_def${DEF}.json <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f frac %.3f verified %s %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"]))
except Exception as e:
    print("ERR", e)
PY
)"
  done; done
done
bash scripts/pmc_any.sh r4c_c2_patch fused_eval_window python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 2 --workload c2_patch > $OUT/pmc_c2_patch.txt 2>&1; cat $OUT/pmc_c2_patch.txt
bash scripts/pmc_any.sh r4c_c4_patch fused_eval_window python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 2 --workload c4_patch > $OUT/pmc_c4_patch.txt 2>&1; cat $OUT/pmc_c4_patch.txt
