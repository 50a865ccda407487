#!/bin/bash
# round 4, session 6 (EXPERIMENTS build): window kernel without the deferred-rows variants; plain loop at 5 / 6 workgroups per CU
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4f; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log | cut -c1-200
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
run() { # tag, env..., workload last
    local tag=$1; shift; local wl=${@: -1}; local envs="${@:1:$#-1}"
    env $envs timeout -k 5 300 $B --workload $wl > $OUT/b_${wl}_$tag.json 2> $OUT/b_${wl}_$tag.err
    echo "$wl $tag: $(python - $OUT/b_${wl}_$tag.json <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f kernel %.3f min %.3f frac %.3f verified %s %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"]))
except Exception as e:
    print("ERR", e)
PY
)"
}
for WL in c2_patch c3_patch c4_patch; do
  run default X=1 $WL
  run plain4 D3F_EXP_WINDOW_PIPE=-1 $WL
  run plain5 D3F_EXP_WINDOW_PIPE=-1 D3F_EXP_WINDOW_OCC=5 $WL
  run plain6 D3F_EXP_WINDOW_PIPE=-1 D3F_EXP_WINDOW_OCC=6 $WL
done
run pipe3 D3F_EXP_WINDOW_OCC=3 c2_patch
run pipe3 D3F_EXP_WINDOW_OCC=3 c4_patch
