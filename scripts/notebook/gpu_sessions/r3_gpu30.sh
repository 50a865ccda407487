#!/bin/bash
# round 3, session 30: the multi-rank code path of bench.py on RCCL with ONE rank (process group, barrier, max over ranks, async all-gather)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_dist1; mkdir -p $OUT
export TMPDIR=/tmp
for G in dist full none; do
  timeout -k 5 300 python bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --gather $G > $OUT/bench_rccl1_$G.json 2> $OUT/bench_rccl1_$G.err; echo "rc=$?"
  tail -c 1500 $OUT/bench_rccl1_$G.json | python -c "
import sys, json
t=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]
if t:
    d=json.loads(t[-1]); c=d['config']
    print(d['value'], d['ms_per_step'], c.get('rccl_world_size'), c.get('backend'), c.get('gather'), c.get('gather_overlap'), c.get('gather_overlap_error'), d.get('verified'))
else: print('no json')
"
  grep -v amdgpu $OUT/bench_rccl1_$G.err | tail -3
done
timeout -k 5 300 python bench.py --gpus 1 --force-dist --workload c5_track --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_rccl1_c5.json 2> $OUT/bench_rccl1_c5.err; echo "rc=$?"; tail -c 300 $OUT/bench_rccl1_c5.json; grep -v amdgpu $OUT/bench_rccl1_c5.err | tail -3
