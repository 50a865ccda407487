#!/bin/bash
# round 5, session 20 (EXPERIMENTS build): the ordering with 8-byte slots, and fewer counting cells (D3F_EXP_ORDER_BITS) -- step time of cloud queries + kernel trace
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s20
V="bits_auto,bits19=D3F_EXP_ORDER_BITS=19,bits18=D3F_EXP_ORDER_BITS=18,bits20=D3F_EXP_ORDER_BITS=20,bits17=D3F_EXP_ORDER_BITS=17"
timeout -k 5 900 python scripts/notebook/exp_cloud.py --out gpurun_out/r5_s20 --variants "$V" --steps 40 --cases c2_patch:random,c3_patch:random,c5_track:random 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s20/log.txt | grep -v '^{' | cut -c1-200
REPO=$(pwd); cd /tmp
for B in 0 19 18; do
D3F_EXP_ORDER_BITS=$B timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r5_s20/t$B/trace -o trace --output-format csv -- python $REPO/bench.py --workload c2_patch --points random --no-cpu-baseline --no-verify --steps 20 > /dev/null 2> $REPO/gpurun_out/r5_s20/t$B.err
echo "bits $B"; grep -E "cell_count|scan_lookback|scatter_kernel|cell_rank|order_clear|window_gate" $REPO/gpurun_out/r5_s20/t$B/trace/trace_kernel_stats.csv | awk -F, '{printf "   %-40s avg %.1f us\n", substr($1,1,40), $4/1000}'
done
