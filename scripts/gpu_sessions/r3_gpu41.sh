#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_phase; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests -m gpu -q -x -k "sliced or c4_dense or bench_workload" 2>&1 | tail -3
timeout -k 5 200 python scripts/exp_phase_timing.py c2_dense > $OUT/phase_c2_dense.txt 2>&1; grep -v amdgpu $OUT/phase_c2_dense.txt
for WL in c2_dense c3_dense c4_dense; do timeout -k 5 300 python bench.py --workload $WL --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print(d['config']['workload'][:10], 'step %.3f kernel %.3f frac %.3f verified %s' % (d['ms_per_step'], r['kernel_ms_avg'], r['frac'], d.get('verified')))"; done
