#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3n; mkdir -p $OUT
export TMPDIR=/tmp
for WL in c2_patch c2_dense c5_track; do
  timeout -k 5 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 1 --backend gloo --workload $WL --no-cpu-baseline > $OUT/two_ranks_$WL.json 2> $OUT/two_ranks_$WL.err
  echo "$WL rc=$?"; tail -1 $OUT/two_ranks_$WL.json | cut -c1-600
done
