#!/bin/bash
# round 3, session 3: stream kernel with TICKETED tile hand-out (smooth front instead of lock-step sweeps)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3e; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "stream_launch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
S="D3F_EXP_STREAM_T"; U="D3F_EXP_STREAM_UNIT"; R="D3F_EXP_STREAM_R"; V="D3F_EXP_STREAM_VAR"; K="D3F_EXP_STREAM_TICKETS=1"; G="D3F_EXP_STREAM_G"
timeout -k 5 900 python scripts/exp_knobs.py c2_dense "old:D3F_EXP_STREAM=-1" \
  "v0g96u64:$K,$V=0,$G=96,$U=64" "v0g96u32:$K,$V=0,$G=96,$U=32" "v0g96u128:$K,$V=0,$G=96,$U=128" "v0g96u256:$K,$V=0,$G=96,$U=256" "v0g64u64:$K,$V=0,$G=64,$U=64" \
  "v3g128u64:$K,$V=3,$G=128,$U=64" "v3g128u32:$K,$V=3,$G=128,$U=32" "v3g128u128:$K,$V=3,$G=128,$U=128" "v3g96u64:$K,$V=3,$G=96,$U=64" \
  "v1g224u64:$K,$V=1,$G=224,$U=64" "v1g224u128:$K,$V=1,$G=224,$U=128" "v1g160u64:$K,$V=1,$G=160,$U=64" "v1g128u64:$K,$V=1,$G=128,$U=64" \
  "v2g160u64:$K,$V=2,$G=160,$U=64" "v2g128u64:$K,$V=2,$G=128,$U=64" "v2g96u64:$K,$V=2,$G=96,$U=64" \
  "T24v0g96u32:$K,$S=24,$V=0,$G=96,$U=32" "T24v1g192u32:$K,$S=24,$V=1,$G=192,$U=32" "T16v1g160u64:$K,$S=16,$V=1,$G=160,$U=64" \
  "L4v0g96u64:$K,D3F_EXP_STREAM_LG=4,$V=0,$G=96,$U=64" "L4v1g224u64:$K,D3F_EXP_STREAM_LG=4,$V=1,$G=224,$U=64" "L4v1g128u64:$K,D3F_EXP_STREAM_LG=4,$V=1,$G=128,$U=64" \
  "old2:D3F_EXP_STREAM=-1" > $OUT/sweep_c2_dense.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep_c2_dense.txt | cut -c1-120
bash scripts/pmc_exp.sh r3e_v0 fused_eval_stream c2_dense "v0:$K,$V=0,$G=96,$U=64" | tee $OUT/pmc_v0.txt | tail -3
bash scripts/pmc_exp.sh r3e_v1 fused_eval_stream c2_dense "v1:$K,$V=1,$G=224,$U=64" | tee $OUT/pmc_v1.txt | tail -3
