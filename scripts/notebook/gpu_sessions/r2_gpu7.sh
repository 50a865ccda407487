#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2g; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "order or reorder or unordered or full_size or walk" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -6 $OUT/pytest.log
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c4_patch c5_track; do timeout -k 5 400 $B --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; done
for WL in c2_dense c2_patch c3_patch; do timeout -k 5 400 $B --workload $WL --points random > $OUT/bench_${WL}_random.json 2> $OUT/bench_${WL}_random.err; done
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_c2pr -o trace --output-format csv -- $B --workload c2_patch --points random --no-verify > /dev/null 2> $OUT/trace.err
cd $REPO
python scripts/summarize_prof.py $OUT/trace_c2pr | head -30
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms | %.3e pts/s | cached-order %.3e | frac %.3f | verified %s | %s" % (d["ms_per_step"], r["kernel_ms_avg"], d["value"], d.get("points_per_s_with_cached_point_order", 0), r["frac"], d.get("verified"), d["config"].get("point_order")))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
