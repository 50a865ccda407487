#!/bin/bash
# round 3, session 4: which side binds the stream kernel?  producer-only / consumer-only timings (results are wrong by design)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3f; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
S="D3F_EXP_STREAM_T"; U="D3F_EXP_STREAM_UNIT"; V="D3F_EXP_STREAM_VAR"; K="D3F_EXP_STREAM_TICKETS=1"; G="D3F_EXP_STREAM_G"; D="D3F_EXP_STREAM_DEBUG"; L="D3F_EXP_STREAM_LG=4"
timeout -k 5 900 python scripts/exp_knobs.py c2_dense "old:D3F_EXP_STREAM=-1" \
  "v2g96:$K,$V=2,$G=96,$U=64" "v2g96 P-only:$K,$V=2,$G=96,$U=64,$D=1" "v2g96 C-only:$K,$V=2,$G=96,$U=64,$D=2" \
  "v1g128:$K,$V=1,$G=128,$U=64" "v1g128 P-only:$K,$V=1,$G=128,$U=64,$D=1" "v1g128 C-only:$K,$V=1,$G=128,$U=64,$D=2" \
  "v1g224 P-only:$K,$V=1,$G=224,$U=64,$D=1" "v1g224 C-only:$K,$V=1,$G=224,$U=64,$D=2" \
  "L4v1g128:$K,$L,$V=1,$G=128,$U=64" "L4v1g128 P-only:$K,$L,$V=1,$G=128,$U=64,$D=1" "L4v1g128 C-only:$K,$L,$V=1,$G=128,$U=64,$D=2" \
  "L4T24v1g128:$K,$L,$S=24,$V=1,$G=128,$U=32" "L4T24v1g128 P-only:$K,$L,$S=24,$V=1,$G=128,$U=32,$D=1" "L4T24v1g128 C-only:$K,$L,$S=24,$V=1,$G=128,$U=32,$D=2" \
  "T24v1g96:$K,$S=24,$V=1,$G=96,$U=32" "T24v1g96 P-only:$K,$S=24,$V=1,$G=96,$U=32,$D=1" \
  > $OUT/sweep.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep.txt | cut -c1-100
