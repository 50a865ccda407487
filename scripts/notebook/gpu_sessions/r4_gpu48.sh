#!/bin/bash
# round 4, session 48 (PRODUCT build, final sources): whole GPU suite + smoke + default bench line
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_v3; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-200
timeout -k 5 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2
timeout -k 5 600 python $REPO/bench.py > $OUT/default_bench.json 2> $OUT/default_bench.err; python - $OUT/default_bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]; c=d["cpu_baseline"]
print("default: value %.4g ms %.3f frac %.3f kernel %.3f verified %s cpu %.3g (%s)" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_ms_avg"], d["verified"], c["value"], c["kind"]))
PY
