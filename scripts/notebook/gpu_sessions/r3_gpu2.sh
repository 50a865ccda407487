#!/bin/bash
# round 3, session 2: stream kernel -- workgroups per unit matched to the resident workgroups, tiles per workgroup; counters
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3b; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
S="D3F_EXP_STREAM_T"; U="D3F_EXP_STREAM_UNIT"; R="D3F_EXP_STREAM_R"; V="D3F_EXP_STREAM_VAR"
timeout -k 5 900 python scripts/exp_knobs.py c2_dense "old:D3F_EXP_STREAM=-1" \
  "v0u96:$V=0,$U=96" "v0u64:$V=0,$U=64" "v0u48:$V=0,$U=48" "v0u128:$V=0,$U=128" "v0u96r4:$V=0,$U=96,$R=4" "v0u96r16:$V=0,$U=96,$R=16" "v0u96r2:$V=0,$U=96,$R=2" \
  "v3u128:$V=3,$U=128" "v3u96:$V=3,$U=96" "v3u64:$V=3,$U=64" "v3u128r16:$V=3,$U=128,$R=16" \
  "v1u224:$V=1,$U=224" "v1u160:$V=1,$U=160" "v1u128:$V=1,$U=128" "v1u96:$V=1,$U=96" \
  "v2u160:$V=2,$U=160" "v2u128:$V=2,$U=128" "v2u96:$V=2,$U=96" \
  "T24v0u96:$S=24,$V=0,$U=96" "T24v0u48:$S=24,$V=0,$U=48" "T24v1u192:$S=24,$V=1,$U=192" "T24v1u96:$S=24,$V=1,$U=96" \
  "L4v0u96:D3F_EXP_STREAM_LG=4,$V=0,$U=96" "L4v1u224:D3F_EXP_STREAM_LG=4,$V=1,$U=224" "L4v1u128:D3F_EXP_STREAM_LG=4,$V=1,$U=128" \
  "old2:D3F_EXP_STREAM=-1" > $OUT/sweep_c2_dense.txt 2>&1
grep -v "^$\|amdgpu.ids" $OUT/sweep_c2_dense.txt | cut -c1-120
bash scripts/pmc_exp.sh r3b_old fused_eval_sliced c2_dense "old:D3F_EXP_STREAM=-1" | tee $OUT/pmc_old.txt
bash scripts/pmc_exp.sh r3b_v0 fused_eval_stream c2_dense "v0:$V=0,$U=96" | tee $OUT/pmc_v0.txt
bash scripts/pmc_exp.sh r3b_v1 fused_eval_stream c2_dense "v1:$V=1,$U=224" | tee $OUT/pmc_v1.txt
