#!/bin/bash
# round 5, session 26 (EXPERIMENTS build): with the box-relative key grid, do fewer counting cells pay?  (scan / clear shrink, atomics and rank loops grow)
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s26
V="bits21,bits20=D3F_EXP_ORDER_BITS=20,bits19=D3F_EXP_ORDER_BITS=19,bits18=D3F_EXP_ORDER_BITS=18"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s26 --variants "$V" --steps 40 --cases c2_patch:random,c3_patch:random,c5_track:random 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s26/log.txt | grep -v '^{' | cut -c1-200
REPO=$(pwd); cd /tmp
for B in 21 20 19 18; do
D3F_EXP_ORDER_BITS=$B timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r5_s26/t$B/trace -o trace --output-format csv -- python $REPO/bench.py --workload c2_patch --points random --no-cpu-baseline --no-verify --steps 20 > /dev/null 2> $REPO/gpurun_out/r5_s26/t$B.err
python - $REPO/gpurun_out/r5_s26/t$B/trace/trace_kernel_stats.csv $B <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("BITS", sys.argv[2], " | ".join("%s %.1f" % (r['Name'].split('(')[0].replace('void d3f::','').replace('d3f::','')[:22], float(r['AverageNs'])/1e3) for r in rows if any(k in r['Name'] for k in ('cell_count','scan_lookback','scatter_kernel','cell_rank','order_clear','order_bbox','fused_eval_window'))))
PY
done
