#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2n; mkdir -p $OUT
export TMPDIR=/tmp
# two gloo ranks on ONE GPU: the N>1 code path of bench.py (sharded query, two async gathers in flight, byte accounting)


timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --no-cpu-baseline --workload c5_track > $OUT/bench_gloo2_c5.json 2> $OUT/bench_gloo2_c5.err; tail -c 600 $OUT/bench_gloo2_c5.json; echo; tail -3 $OUT/bench_gloo2_c5.err

