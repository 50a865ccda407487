#!/bin/bash
# round 3, session 6: four-stage producer pipeline (no exposed producer load)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3h; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "stream_launch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
S="D3F_EXP_STREAM_T"; U="D3F_EXP_STREAM_UNIT"; V="D3F_EXP_STREAM_VAR"; K="D3F_EXP_STREAM_TICKETS=1"; G="D3F_EXP_STREAM_G"; D="D3F_EXP_STREAM_DEBUG"; L="D3F_EXP_STREAM_LG=4"
timeout -k 5 900 python scripts/exp_knobs.py c2_dense "old:D3F_EXP_STREAM=-1" \
  "v2g96:$K,$V=2,$G=96,$U=64" "v2g96 P-only:$K,$V=2,$G=96,$U=64,$D=1" "v2g96 C-only:$K,$V=2,$G=96,$U=64,$D=2" \
  "v2g128:$K,$V=2,$G=128,$U=64" "v2g160:$K,$V=2,$G=160,$U=64" "v2g80:$K,$V=2,$G=80,$U=64" \
  "v0g96:$K,$V=0,$G=96,$U=64" "v0g64:$K,$V=0,$G=64,$U=64" "v0g80:$K,$V=0,$G=80,$U=64" "v3g96:$K,$V=3,$G=96,$U=64" "v3g128:$K,$V=3,$G=128,$U=64" \
  "v1g128:$K,$V=1,$G=128,$U=64" "v1g160:$K,$V=1,$G=160,$U=64" "v1g224:$K,$V=1,$G=224,$U=64" \
  "L4v1g128:$K,$L,$V=1,$G=128,$U=64" "L4v1g128 P-only:$K,$L,$V=1,$G=128,$U=64,$D=1" "L4v1g160:$K,$L,$V=1,$G=160,$U=64" "L4v1g224:$K,$L,$V=1,$G=224,$U=64" "L4v0g96:$K,$L,$V=0,$G=96,$U=64" \
  "L4T24v1g128:$K,$L,$S=24,$V=1,$G=128,$U=32" "L4T24v1g224:$K,$L,$S=24,$V=1,$G=224,$U=32" "L4T24v0g96:$K,$L,$S=24,$V=0,$G=96,$U=32" \
  "T24v2g96:$K,$S=24,$V=2,$G=96,$U=32" "T16v2g96:$K,$S=16,$V=2,$G=96,$U=64" "T16v1g128:$K,$S=16,$V=1,$G=128,$U=64" \
  "static v2u96:$V=2,$U=96" "old2:D3F_EXP_STREAM=-1" \
  > $OUT/sweep.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep.txt | cut -c1-100
