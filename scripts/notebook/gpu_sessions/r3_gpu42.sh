#!/bin/bash
set -u
for WL in c2_dense c2_dense c3_dense c4_dense; do timeout -k 5 300 python bench.py --workload $WL --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print(d['config']['workload'][:10], 'step %.3f kernel %.3f min %.3f frac %.3f verified %s' % (d['ms_per_step'], r['kernel_ms_avg'], r['kernel_ms_min'], r['frac'], d.get('verified')))"; done
