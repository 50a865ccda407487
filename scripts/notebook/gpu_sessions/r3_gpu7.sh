#!/bin/bash
# round 3, session 7: geometry pre-pass (phase A once per point) -> finer channel slices without the repeated phase A
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3i; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "stream_launch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
S="D3F_EXP_STREAM_T"; U="D3F_EXP_STREAM_UNIT"; V="D3F_EXP_STREAM_VAR"; K="D3F_EXP_STREAM_TICKETS=1"; G="D3F_EXP_STREAM_G"; D="D3F_EXP_STREAM_DEBUG"; L="D3F_EXP_STREAM_LG"; P="D3F_EXP_STREAM_PRE=1"
timeout -k 5 900 python scripts/exp_knobs.py c2_dense "old:D3F_EXP_STREAM=-1" \
  "v2g96:$K,$V=2,$G=96,$U=64" \
  "pre L5 v2g96:$K,$P,$V=2,$G=96,$U=64" "pre L5 v1g128:$K,$P,$V=1,$G=128,$U=64" "pre L5 v2g96 P-only:$K,$P,$V=2,$G=96,$U=64,$D=1" \
  "pre L4T12 v1g128:$K,$P,$L=4,$V=1,$G=128,$U=64" "pre L4T12 v1g160:$K,$P,$L=4,$V=1,$G=160,$U=64" "pre L4T12 v1g224:$K,$P,$L=4,$V=1,$G=224,$U=64" \
  "pre L4T24 v1g96:$K,$P,$L=4,$S=24,$V=1,$G=96,$U=32" "pre L4T24 v1g128:$K,$P,$L=4,$S=24,$V=1,$G=128,$U=32" "pre L4T24 v1g160:$K,$P,$L=4,$S=24,$V=1,$G=160,$U=32" \
  "pre L4T24 v2g96:$K,$P,$L=4,$S=24,$V=2,$G=96,$U=32" "pre L4T24 v2g128:$K,$P,$L=4,$S=24,$V=2,$G=128,$U=32" "pre L4T24 v0g96:$K,$P,$L=4,$S=24,$V=0,$G=96,$U=32" "pre L4T24 v0g64:$K,$P,$L=4,$S=24,$V=0,$G=64,$U=32" \
  "pre L3T24 v1g128:$K,$P,$L=3,$S=24,$V=1,$G=128,$U=32" "pre L3T24 v1g160:$K,$P,$L=3,$S=24,$V=1,$G=160,$U=32" "pre L3T24 v1g224:$K,$P,$L=3,$S=24,$V=1,$G=224,$U=32" \
  "pre L3T24 v2g96:$K,$P,$L=3,$S=24,$V=2,$G=96,$U=32" "pre L3T24 v2g128:$K,$P,$L=3,$S=24,$V=2,$G=128,$U=32" "pre L3T24 v2g160:$K,$P,$L=3,$S=24,$V=2,$G=160,$U=32" \
  "pre L4T24 v1g128 P-only:$K,$P,$L=4,$S=24,$V=1,$G=128,$U=32,$D=1" "pre L3T24 v1g160 P-only:$K,$P,$L=3,$S=24,$V=1,$G=160,$U=32,$D=1" \
  "old2:D3F_EXP_STREAM=-1" \
  > $OUT/sweep.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep.txt | cut -c1-108
