#!/bin/bash
# round 4, session 31 (EXPERIMENTS build): the sliced launch's geometry again, now that the rows no longer occupy the L2s:
# workgroups per unit (points per unit = 16 x that), tile size, slice width, interleave, views in flight -- C2-dense
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4ad; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
B="python $REPO/bench.py --no-cpu-baseline --steps 20 --workload c2_dense"
line() { python - $1 <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("kernel %.3f min %.3f frac %.3f verified %s" % (r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print("ERR", e)
PY
}
run() { env "$@" timeout -k 5 300 $B > $OUT/x.json 2> $OUT/x.err; echo "$*: $(line $OUT/x.json)"; }
run D3F_EXP_NONE=0
for U in 64 128 192 384 512 768 1024 2048; do run D3F_EXP_SLICED_UNIT=$U; done
for T in 8 32 64; do run D3F_EXP_SLICED_TILE=$T; done
run D3F_EXP_SLICED_TILE=32 D3F_EXP_SLICED_UNIT=256
run D3F_EXP_SLICED_TILE=32 D3F_EXP_SLICED_UNIT=512
run D3F_EXP_SLICED_ILV=2
run D3F_EXP_SLICED_ILV=2 D3F_EXP_SLICED_UNIT=512
run D3F_EXP_SLICED_VC=4
run D3F_EXP_SLICED_VC=1
run D3F_EXP_SLICED=2
run D3F_EXP_SLICED=-1
run D3F_EXP_NONE=0
