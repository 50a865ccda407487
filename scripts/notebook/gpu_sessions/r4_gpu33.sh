#!/bin/bash
# round 4, session 33 (PRODUCT build): the N > 1 code path on RCCL with ONE rank (all a one-GPU box allows): `--gpus 1 --force-dist`
# for the N > 1 default workload (c4_patch), with the per-point gather the timed step uses and with the full-field gather
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_v3/rccl_one_rank; mkdir -p $OUT
export TMPDIR=/tmp
for G in dist full; do
  timeout -k 5 400 python $REPO/bench.py --gpus 1 --force-dist --workload c4_patch --gather $G --no-cpu-baseline > $OUT/c4_patch_gather_$G.json 2> $OUT/c4_patch_gather_$G.err; echo "rc=$?"
  python - $OUT/c4_patch_gather_$G.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d["config"]
    print("step %.3f value %.3g verified %s backend %s world %s gather %s ipc_retry %s" % (d["ms_per_step"], d["value"], d.get("verified"), c.get("backend"), c.get("rccl_world_size"), c.get("gather"), c.get("ipc_mode_retry")))
except Exception as e:
    print("ERR", e)
PY
done
timeout -k 5 400 python $REPO/bench.py --gpus 1 --force-dist --no-cpu-baseline > $OUT/default_workload.json 2> $OUT/default_workload.err; echo "rc=$?"; tail -c 200 $OUT/default_workload.json
