#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3o; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q -x -k "corr or pairwise or softmax or similarity or knn or sharding or config_5 or driver" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
timeout -k 5 300 $B --workload c5_track > $OUT/bench_c5_track.json 2>$OUT/bench_c5_track.err
cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/c5prof -o trace --output-format csv -- $B --workload c5_track --no-verify > /dev/null 2>&1; cd $REPO
python scripts/summarize_prof.py $OUT/c5prof 2>/dev/null | grep -i "softmax\|pairwise" | head -8
python - "$OUT/bench_c5_track.json" <<'PY'
import json,sys
t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
print("c5 step %.3f ms | %.3e pts/s | verified %s" % (d["ms_per_step"], d["value"], d.get("verified")))
PY
