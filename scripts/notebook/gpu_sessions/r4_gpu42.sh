#!/bin/bash
# round 4, session 42 (PRODUCT build): the N > 1 launch path end to end on one box: `bench.py --gpus 2 --backend gloo` (two ranks sharing
# the GPU; default workload for N > 1 = c4_patch), and the two-rank sharding test on the HIP kernels
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_v3/two_ranks_one_gpu; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python $REPO/bench.py --gpus 2 --backend gloo --no-cpu-baseline > $OUT/bench_gpus2_gloo.json 2> $OUT/bench_gpus2_gloo.err; echo "rc=$?"
python - $OUT/bench_gpus2_gloo.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d["config"]
    print("n_gpus %s step %.3f value %.3g verified %s backend %s world %s workload %s gather %s overlap %s" % (d["n_gpus"], d["ms_per_step"], d["value"], d.get("verified"), c.get("backend"), c.get("rccl_world_size"), c["workload"][:40], c.get("gather"), c.get("gather_overlap")))
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
timeout -k 5 600 python -m pytest tests/test_gpu_sharding.py -m gpu -q 2>&1 | tail -2
