#!/bin/bash
# round 3, session 23: channel-sliced launch on Morton-ordered clouds (dense maps); probe kernel with 1024 samples
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_cloud; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q -x -k "sliced or order or probe or walks or bench_workload" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in c2_dense c4_dense c3_dense c5_track; do
  timeout -k 5 600 $B --workload $WL --points random > $OUT/bench_${WL}_random.json 2> $OUT/bench_${WL}_random.err
  python - $OUT/bench_${WL}_random.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
    print(d["config"]["workload"][:30], "| step %.3f ms | kernel %.3f ms | frac %.3f | verified %s | %s | %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["frac"], d.get("verified"), r["kernel"], d["config"].get("point_order")))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-400:], open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
done
cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/c5_trace -o trace --output-format csv -- python $REPO/bench.py --workload c5_track --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $OUT/c5_under_rocprof.json 2> $OUT/c5_trace.err; cd $REPO
python scripts/kernel_stats.py $OUT/c5_trace d3f:: > $OUT/c5_kernel_stats.txt; head -8 $OUT/c5_kernel_stats.txt
rm -rf $OUT/*/trace/*/*hip_api* 2>/dev/null; du -sh $OUT
