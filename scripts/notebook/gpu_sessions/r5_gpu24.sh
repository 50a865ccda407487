#!/bin/bash
# round 5, session 24 (EXPERIMENTS build): key grid laid over the cloud's bounding box vs the fixed 4-mm grid -- step time of cloud queries + kernel trace + exact-order test
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s24
V="bbox,fixed=D3F_EXP_ORDER_FIXED_GRID=1"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s24 --variants "$V" --steps 40 --cases c2_patch:random,c3_patch:random 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s24/log.txt | grep -v '^{' | cut -c1-200
grep oracle gpurun_out/r5_s24/log.txt | cut -c1-160
timeout -k 5 300 python -m pytest tests/test_gpu_walks.py -q -x -k "hilbert_order or cloud_gate or window" 2>&1 | tail -5
REPO=$(pwd); cd /tmp
for X in 0 1; do
D3F_EXP_ORDER_FIXED_GRID=$X timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r5_s24/t$X/trace -o trace --output-format csv -- python $REPO/bench.py --workload c2_patch --points random --no-cpu-baseline --steps 20 > $REPO/gpurun_out/r5_s24/b$X.json 2> $REPO/gpurun_out/r5_s24/t$X.err
python - $REPO/gpurun_out/r5_s24/t$X/trace/trace_kernel_stats.csv $X <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("FIXED_GRID", sys.argv[2], " | ".join("%s %.1f" % (r['Name'].split('(')[0].replace('void d3f::','').replace('d3f::','')[:22], float(r['AverageNs'])/1e3) for r in rows if any(k in r['Name'] for k in ('cell_count','scan_lookback','scatter_kernel','cell_rank','order_clear','order_bbox','gate_probe','fused_eval'))))
PY
grep -o '"verified": [a-z]*' $REPO/gpurun_out/r5_s24/b$X.json
done
