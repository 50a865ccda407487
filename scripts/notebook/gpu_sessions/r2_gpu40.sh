#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3k; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x -k "cell_run or random_conf or config4 or reorder or fast_path or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c4_patch c5_track; do timeout -k 5 300 $B --workload $WL > $OUT/bench_$WL.json 2>$OUT/bench_$WL.err; done
for WL in c2_patch c3_patch; do timeout -k 5 300 $B --workload $WL --points random > $OUT/bench_${WL}_random.json 2>$OUT/bench_${WL}_random.err; done
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
