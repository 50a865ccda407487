#!/bin/bash
# round 3, session 5: counters of the consumer-only run (stale records: every load misses the L2 but lives in the Infinity Cache)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3g; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
U="D3F_EXP_STREAM_UNIT"; V="D3F_EXP_STREAM_VAR"; K="D3F_EXP_STREAM_TICKETS=1"; G="D3F_EXP_STREAM_G"; D="D3F_EXP_STREAM_DEBUG"
bash scripts/pmc_exp.sh r3g_conly fused_eval_stream c2_dense "c:$K,$V=2,$G=96,$U=64,$D=2" | tee $OUT/pmc_conly.txt
bash scripts/pmc_exp.sh r3g_v2 fused_eval_stream c2_dense "c:$K,$V=2,$G=96,$U=64" | tee $OUT/pmc_v2.txt
