#!/bin/bash
# round 5, session 27 (EXPERIMENTS build): key / histogram kernel with four points per lane (loads, then atomics, in flight together) vs one
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s27
V="ppt4,ppt1=D3F_EXP_ORDER_PPT1=1"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s27 --variants "$V" --steps 40 --cases c2_patch:random,c3_patch:random,c5_track:random 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s27/log.txt | grep -v '^{' | cut -c1-200
REPO=$(pwd); cd /tmp
for B in 0 1; do
D3F_EXP_ORDER_PPT1=$B timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r5_s27/t$B/trace -o trace --output-format csv -- python $REPO/bench.py --workload c2_patch --points random --no-cpu-baseline --no-verify --steps 20 > /dev/null 2> $REPO/gpurun_out/r5_s27/t$B.err
python - $REPO/gpurun_out/r5_s27/t$B/trace/trace_kernel_stats.csv $B <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("PPT1", sys.argv[2], " | ".join("%s %.1f" % (r['Name'].split('(')[0].replace('void d3f::','').replace('d3f::','')[:22], float(r['AverageNs'])/1e3) for r in rows if any(k in r['Name'] for k in ('cell_count','scan_lookback','scatter_kernel','cell_rank','order_clear','order_bbox','fused_eval_window'))))
PY
done
cd $REPO; timeout -k 5 300 python -m pytest tests/test_gpu_walks.py -q -x -k "hilbert_order" 2>&1 | tail -2
