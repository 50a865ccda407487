#!/bin/bash
# round 6, session 57 (PRODUCT build, FINAL sources of round 6: with the distance-only kernel): GPU suite, rocprofv3 evidence of every workload (scripts/r6_profile_all.sh -> r6_v4),
# one verified bench line per workload, the default bench line
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s57; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
scripts/r6_profile_all.sh r6_v4 > $OUT/profile_all.log 2>&1
tail -2 $OUT/profile_all.log
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-30s value %.4g step %.4f ms kernel avg %.4f min %.4f frac %.3f traffic %s in-run %s verified %s  %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], r["traffic"], r["traffic_measured_in_run"], d.get("verified"), r["kernel"][:58]))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
for SPEC in c2_dense c3_dense c2_patch c3_patch c4_patch ref_patch c2_patch:random c3_patch:random ref_patch:random ref_patch:surface c4_patch:random c2_dense:random c5_track c2_patch_f16 c2_dense_f16 dist_only c4_dense; do
  WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
  timeout -k 5 400 python bench.py --no-cpu-baseline --steps 30 --workload $WL --points $PTS > $OUT/${WL}_$PTS.json 2> $OUT/${WL}_$PTS.err
  line $OUT/${WL}_$PTS.json "$WL $PTS"
done
mkdir -p $OUT/rccl_one_rank $OUT/two_ranks_one_gpu
timeout -k 5 400 python bench.py --gpus 1 --force-dist --steps 20 > $OUT/rccl_one_rank/bench_line.json 2> $OUT/rccl_one_rank/bench_line.err; line $OUT/rccl_one_rank/bench_line.json "rccl one rank"
timeout -k 5 400 python bench.py --gpus 2 --backend gloo --steps 10 --traffic off > $OUT/two_ranks_one_gpu/bench_line.json 2> $OUT/two_ranks_one_gpu/bench_line.err; line $OUT/two_ranks_one_gpu/bench_line.json "two gloo ranks, one GPU"
timeout -k 5 600 python bench.py > $OUT/default_bench_line.json 2> $OUT/default_bench_line.err; line $OUT/default_bench_line.json "default"
