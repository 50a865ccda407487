#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3k; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
U="D3F_EXP_STREAM_UNIT"; V="D3F_EXP_STREAM_VAR"; K="D3F_EXP_STREAM_TICKETS=1"; G="D3F_EXP_STREAM_G"; P="D3F_EXP_STREAM_PRE=1"
timeout -k 5 900 python scripts/exp_knobs.py c3_dense "old:D3F_EXP_STREAM=-1" "v2g96:$K,$V=2,$G=96,$U=64" "v1g128:$K,$V=1,$G=128,$U=64" "v2g112:$K,$V=2,$G=112,$U=64" "v1g112:$K,$V=1,$G=112,$U=64" "v0g96:$K,$V=0,$G=96,$U=64" "old2:D3F_EXP_STREAM=-1" > $OUT/sweep3.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep3.txt | cut -c1-150
timeout -k 5 900 python scripts/exp_knobs.py c2_dense "old:D3F_EXP_STREAM=-1" "v2g104:$K,$V=2,$G=104,$U=64" "v2g112:$K,$V=2,$G=112,$U=64" "v1g112:$K,$V=1,$G=112,$U=64" "v1g144:$K,$V=1,$G=144,$U=64" "v1g128u128:$K,$V=1,$G=128,$U=128" "v1g128u32:$K,$V=1,$G=128,$U=32" "v2g96u128:$K,$V=2,$G=96,$U=128" "old2:D3F_EXP_STREAM=-1" > $OUT/sweep2.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep2.txt | cut -c1-150
