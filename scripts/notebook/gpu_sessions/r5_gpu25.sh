#!/bin/bash
# round 5, session 25 (PRODUCT build): untraced bench lines of the cloud workloads after the box-relative key grid; smoke(); default bench
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_s25
for W in "c2_patch random" "c3_patch random" "ref_patch random" "ref_patch surface" "c4_patch random" "c2_dense random" "c5_track random"; do
set -- $W
timeout -k 5 300 python bench.py --workload $1 --points $2 --no-cpu-baseline --steps 40 2>/dev/null > gpurun_out/r5_s25/$1_$2.json
python - gpurun_out/r5_s25/$1_$2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("%-10s %-8s value %.3e  ms/step %.4f  kernel_ms %s frac %.3f verified %s gate %s" % (d["config"]["workload"][:10], d["config"].get("points"), d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms_avg"), d["roofline"]["frac"], d.get("verified"), d["config"].get("device_gate")))
PY
done
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 5 600 python bench.py 2>/dev/null | tee gpurun_out/r5_s25/default.json | cut -c1-600
