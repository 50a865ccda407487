#!/bin/bash
# round 4, session 13 (EXPERIMENTS build): two-half pool schedule of the window kernel: tests, bench lines, phase stamps
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4m; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log | cut -c1-200
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c2_patch c3_patch c4_patch ref_patch; do
    timeout -k 5 300 $B --workload $WL > $OUT/b_${WL}.json 2> $OUT/b_${WL}.err
    echo "$WL: $(python - $OUT/b_${WL}.json <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f frac %.3f verified %s %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"]))
except Exception as e:
    print("ERR", e)
PY
)"
done
D3F_EXP_STAMPS=1 timeout -k 5 300 python scripts/exp_stamps.py c2_patch c4_patch > $OUT/stamps.txt 2>&1; grep -v amdgpu $OUT/stamps.txt
