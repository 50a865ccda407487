#!/bin/bash
# round 6, session 1 (PRODUCT build = round 5's final sources): (a) the clean LDS read-rate probe (lds_read_rate.hip: 16 independent
# reads per s_waitcnt, EXEC masks along the hardware lane groups and along contiguous quarters, 1-8 waves per SIMD); (b) SQ / LDS
# counter passes of every fused kernel family, vector-memory passes for the cell-run and dense kernels
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s1; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && hipcc --offload-arch=gfx950 -O3 $REPO/scripts/notebook/microbench/lds_read_rate.hip -o /tmp/lds_read_rate 2>/dev/null && timeout -k 5 300 /tmp/lds_read_rate) > $OUT/lds_read_rate.txt 2>&1
tail -5 $OUT/lds_read_rate.txt
timeout -k 5 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
scripts/r6_counters.sh r6_s1 c2_patch c3_patch ref_patch c4_patch c2_patch:random ref_patch:surface c2_patch_f16 dist_only c5_track > /dev/null 2>&1
TCP=1 scripts/r6_counters.sh r6_s1 c4_patch:random c2_dense c3_dense c2_dense_f16 > /dev/null 2>&1
ls $OUT
