#!/bin/bash
# round 3, session 1: the persistent producer / consumer kernel (fuse_stream.hip): bit-identity, then the variant sweep
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3a; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "stream_launch or channel_sliced" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -15 $OUT/pytest.log
S="D3F_EXP_STREAM_T"
timeout -k 5 600 python scripts/exp_knobs.py c2_dense "old:D3F_EXP_STREAM=-1" \
  "T12v0:$S=12,D3F_EXP_STREAM_VAR=0" "T12v3:$S=12,D3F_EXP_STREAM_VAR=3" "T12v1:$S=12,D3F_EXP_STREAM_VAR=1" "T12v2:$S=12,D3F_EXP_STREAM_VAR=2" \
  "T24v0:$S=24,D3F_EXP_STREAM_VAR=0" "T24v1:$S=24,D3F_EXP_STREAM_VAR=1" "T24v2:$S=24,D3F_EXP_STREAM_VAR=2" \
  "T16v0:$S=16,D3F_EXP_STREAM_VAR=0" "T16v1:$S=16,D3F_EXP_STREAM_VAR=1" "T16v2:$S=16,D3F_EXP_STREAM_VAR=2" \
  "L4T12v0:D3F_EXP_STREAM_LG=4,$S=12,D3F_EXP_STREAM_VAR=0" "L4T12v1:D3F_EXP_STREAM_LG=4,$S=12,D3F_EXP_STREAM_VAR=1" \
  "L4T24v0:D3F_EXP_STREAM_LG=4,$S=24,D3F_EXP_STREAM_VAR=0" "L4T24v1:D3F_EXP_STREAM_LG=4,$S=24,D3F_EXP_STREAM_VAR=1" \
  "old2:D3F_EXP_STREAM=-1" > $OUT/sweep_c2_dense.txt 2>&1
cat $OUT/sweep_c2_dense.txt | grep -v "^$" | tail -20
timeout -k 5 300 python scripts/exp_knobs.py c3_dense "old:D3F_EXP_STREAM=-1" "T12v1:$S=12,D3F_EXP_STREAM_VAR=1" "T12v0:$S=12,D3F_EXP_STREAM_VAR=0" "T24v1:$S=24,D3F_EXP_STREAM_VAR=1" "T16v1:$S=16,D3F_EXP_STREAM_VAR=1" > $OUT/sweep_c3_dense.txt 2>&1
tail -6 $OUT/sweep_c3_dense.txt
