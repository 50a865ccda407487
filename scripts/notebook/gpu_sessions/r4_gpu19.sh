#!/bin/bash
# round 4, session 19 (EXPERIMENTS build): split launch of the window kernel (set-up kernel -> images in a buffer -> gather kernel),
# prototype behind D3F_EXP_WINDOW_SPLIT=1: equality tests with it on, then bench lines with and without
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4s; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
D3F_EXP_WINDOW_SPLIT=1 timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest (split) rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-200
B="python $REPO/bench.py --no-cpu-baseline --steps 30"
for ROUND in 1 2; do
for SPLIT in 0 1; do
  for WL in c2_patch c3_patch c4_patch ref_patch; do
    D3F_EXP_WINDOW_SPLIT=$SPLIT timeout -k 5 300 $B --workload $WL > $OUT/b_${SPLIT}_${WL}_$ROUND.json 2> $OUT/b_${SPLIT}_${WL}_$ROUND.err
    echo "split $SPLIT $WL: $(python - $OUT/b_${SPLIT}_${WL}_$ROUND.json <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f frac %.3f verified %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print("ERR", e)
PY
)"
  done
done
done
cd /tmp && D3F_EXP_WINDOW_SPLIT=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c2p -o c2p -- python $REPO/bench.py --no-cpu-baseline --steps 20 --workload c2_patch > $OUT/prof_c2p.log 2>&1
python $REPO/scripts/summarize_prof.py $OUT/prof_c2p 2>/dev/null | head -12
