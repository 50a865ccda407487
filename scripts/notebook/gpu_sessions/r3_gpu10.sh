#!/bin/bash
# round 3, session 10: counters of the geometry-pre-pass variants with finer slices (do the fills drop? why no time gain?)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3l; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
S="D3F_EXP_STREAM_T"; U="D3F_EXP_STREAM_UNIT"; V="D3F_EXP_STREAM_VAR"; K="D3F_EXP_STREAM_TICKETS=1"; G="D3F_EXP_STREAM_G"; L="D3F_EXP_STREAM_LG"; P="D3F_EXP_STREAM_PRE=1"
bash scripts/pmc_exp.sh r3l_L5 fused_eval_stream c2_dense "c:$K,$P,$V=1,$G=128,$U=64" | tee $OUT/pmc_pre_L5_v1g128.txt | tail -12
bash scripts/pmc_exp.sh r3l_L4 fused_eval_stream c2_dense "c:$K,$P,$L=4,$V=1,$G=160,$U=64" | tee $OUT/pmc_pre_L4T12_v1g160.txt | tail -12
bash scripts/pmc_exp.sh r3l_L3 fused_eval_stream c2_dense "c:$K,$P,$L=3,$S=24,$V=1,$G=160,$U=32" | tee $OUT/pmc_pre_L3T24_v1g160.txt | tail -12
