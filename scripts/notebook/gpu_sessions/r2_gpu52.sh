#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3v; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>$OUT/bench_${WL}_$TAG.err; }
S="D3F_EXP_SLICED=3 D3F_EXP_SLICED_VC=2"
run base c2_dense D3F_EXP_SLICED=0
run sl16 c2_dense $S D3F_EXP_SLICED_TILE=16
run sl8 c2_dense $S D3F_EXP_SLICED_TILE=8
run sl16p28 c2_dense $S D3F_EXP_SLICED_TILE=16 D3F_EXP_SLICED_PAD=28
run sl16p36 c2_dense $S D3F_EXP_SLICED_TILE=16 D3F_EXP_SLICED_PAD=36
run sl16p50 c2_dense $S D3F_EXP_SLICED_TILE=16 D3F_EXP_SLICED_PAD=50
run sl32p28 c2_dense $S D3F_EXP_SLICED_PAD=28
run sl32p36 c2_dense $S D3F_EXP_SLICED_PAD=36
run sl16vc4 c2_dense D3F_EXP_SLICED=3 D3F_EXP_SLICED_VC=4 D3F_EXP_SLICED_TILE=16
run sl16lg4 c2_dense D3F_EXP_SLICED=2 D3F_EXP_SLICED_VC=2 D3F_EXP_SLICED_TILE=16
run sl32p28 c3_dense D3F_EXP_SLICED_PAD=28
run sl32p36 c3_dense D3F_EXP_SLICED_PAD=36
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms (min %.3f) | frac %.3f | verified %s | %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
