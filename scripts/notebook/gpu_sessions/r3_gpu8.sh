#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3j; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
U="D3F_EXP_STREAM_UNIT"; V="D3F_EXP_STREAM_VAR"; K="D3F_EXP_STREAM_TICKETS=1"; G="D3F_EXP_STREAM_G"; P="D3F_EXP_STREAM_PRE=1"
timeout -k 5 900 python scripts/exp_knobs.py c2_dense "old:D3F_EXP_STREAM=-1" "static v2:$V=2" "v2g96:$K,$V=2,$G=96,$U=64" "v2g96 again:$K,$V=2,$G=96,$U=64" "v1g128:$K,$V=1,$G=128,$U=64" "pre v2g96:$K,$P,$V=2,$G=96,$U=64" "v2g7u5:$K,$V=2,$G=7,$U=5" > $OUT/sweep.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep.txt | cut -c1-250
