#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_l2probe; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/microbench/l2_bw.hip -o /tmp/l2_bw 2> /dev/null
timeout -k 5 120 /tmp/l2_bw > $OUT/l2_bw.txt 2>&1; cat $OUT/l2_bw.txt
