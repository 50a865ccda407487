#!/bin/bash
# final check of the round: full GPU suite, smoke, the default bench line
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2_final4; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 5 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-250
python - "$OUT/bench_default.json" <<'PY'
import json,sys
t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
print("value %.4e | step %.3f | kernel %.3f | frac %.3f | verified %s | cpu %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["verified"], d["cpu_baseline"]["value"]))
PY
