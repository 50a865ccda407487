#!/bin/bash
# round 4, session 39 (EXPERIMENTS build): C3-dense (wide map + 8-instance mask riding along) on the sliced launch's 16-point tiles
# instead of the 32-point ones it gets because of the thin map; units of 256 / 512 workgroups
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ai; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
B="python $REPO/bench.py --no-cpu-baseline --steps 20 --workload c3_dense"
line() { python - $1 <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("kernel %.3f min %.3f frac %.3f verified %s tile %s" % (r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), d["config"].get("tile_points")))
except Exception as e:
    print("ERR", e)
PY
}
run() { env "$@" timeout -k 5 300 $B > $OUT/x.json 2> $OUT/x.err; echo "$*: $(line $OUT/x.json)"; tail -2 $OUT/x.err | grep -v amdgpu | cut -c1-200; }
run D3F_EXP_NONE=0
run D3F_EXP_SLICED_TILE=16
run D3F_EXP_SLICED_TILE=16 D3F_EXP_SLICED_UNIT=512
run D3F_EXP_SLICED_UNIT=256
run D3F_EXP_SLICED_UNIT=64
run D3F_EXP_NONE=0
