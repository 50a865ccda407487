#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2y; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walks.py -m gpu -q -x -k "order or reorder or unordered or probe" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c5_track c2_patch c4_patch; do timeout -k 5 300 $B --workload $WL > $OUT/bench_$WL.json 2>$OUT/bench_$WL.err; done
timeout -k 5 300 $B --workload c2_patch --points random > $OUT/bench_c2_patch_random.json 2>$OUT/bench_c2_patch_random.err
cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/c5prof -o trace --output-format csv -- $B --workload c5_track --no-verify > /dev/null 2>&1; cd $REPO
python scripts/summarize_prof.py $OUT/c5prof 2>/dev/null | grep -i "locality\|scan\|morton\|window_rank\|clear\|scatter\|lattice" | head -12
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms (min %.3f) | %.3e pts/s | verified %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], d["value"], d.get("verified")))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
