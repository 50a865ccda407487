#!/bin/bash
# round 4, session 40 (PRODUCT build): __graft_entry__.smoke(), the examples, and the default bench line with the final library
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_v3; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -3
timeout -k 5 300 python examples/track_synthetic.py 2>&1 | grep -v amdgpu | tail -3
timeout -k 5 600 python $REPO/bench.py > $OUT/default_bench.json 2> $OUT/default_bench.err; python - $OUT/default_bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]; c=d["cpu_baseline"]
print("default: value %.4g ms %.3f frac %.3f kernel %.3f verified %s cpu %.3g (%s)" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_ms_avg"], d["verified"], c["value"], c["kind"]))
PY
