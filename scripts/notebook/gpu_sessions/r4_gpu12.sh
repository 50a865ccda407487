#!/bin/bash
# round 4, session 12 (PRODUCT build): whole GPU suite, smoke(), every bench line (default with the CPU baseline), dist_only profile
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4l; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log | cut -c1-200
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 5 600 python bench.py > $OUT/default_bench.json 2> $OUT/default_bench.err; tail -c 1200 $OUT/default_bench.json; echo
bash scripts/r3_bench_all.sh r4l/bench | tail -16 | cut -c1-170
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
for WL in ref_patch dist_only; do timeout -k 5 400 $B --workload $WL > $OUT/bench/bench_$WL.json 2> $OUT/bench/bench_$WL.err; done
for f in $OUT/bench/bench_ref_patch.json $OUT/bench/bench_dist_only.json; do python - $f <<'PY'
import json,sys
t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
print("%s: step %.3f kernel %.3f pts/s %.3e frac %.3f valu %s verified %s %s" % (sys.argv[1].split('/')[-1], d["ms_per_step"], r["kernel_ms_avg"], d["value"], r["frac"], (r.get("valu_issue") or {}).get("frac_of_issue_peak"), d.get("verified"), r["kernel"]))
PY
done
bash scripts/r4_profile_all.sh r4_v2 dist_only c2_patch c4_patch | tail -3
