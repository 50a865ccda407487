#!/bin/bash
# round 4, session 14 (EXPERIMENTS build): does the pipelined window kernel now win on Morton-ordered CLOUDS (it lost in round 2)?
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4n; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
run() { local tag=$1; shift; local wl=${@: -1}; local envs="${@:1:$#-1}"
    env $envs timeout -k 5 300 $B --workload $wl $PTS > $OUT/b_${wl}_$tag.json 2> $OUT/b_${wl}_$tag.err
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
PTS="--points random"
for WL in c2_patch c3_patch c4_patch; do
  run runs X=1 $WL
  run window D3F_EXP_WINDOW=64 $WL
done
PTS=""
run runs X=1 c5_track
run window D3F_EXP_WINDOW=64 c5_track
