#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3w; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>$OUT/bench_${WL}_$TAG.err; }
S="D3F_EXP_SLICED=3 D3F_EXP_SLICED_TILE=16"
run base1 c2_dense D3F_EXP_SLICED=0
run sl16a c2_dense $S D3F_EXP_SLICED_VC=2
run base2 c2_dense D3F_EXP_SLICED=0
run sl16b c2_dense $S D3F_EXP_SLICED_VC=2
run sl16u128 c2_dense $S D3F_EXP_SLICED_VC=2 D3F_EXP_SLICED_UNIT=128
run sl16u1024 c2_dense $S D3F_EXP_SLICED_VC=2 D3F_EXP_SLICED_UNIT=1024
run sl16u64 c2_dense $S D3F_EXP_SLICED_VC=2 D3F_EXP_SLICED_UNIT=64
run sl16p19 c2_dense $S D3F_EXP_SLICED_VC=2 D3F_EXP_SLICED_PAD=19
run sl16vc1 c2_dense $S D3F_EXP_SLICED_VC=1
run sl16plain c2_dense $S D3F_EXP_SLICED_VC=2 D3F_EXP_STORE=-1
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
