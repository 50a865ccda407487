#!/bin/bash
# round 4, session 11 (EXPERIMENTS build): the view-pair experiment on C2-dense (kernel times only; orders precomputed with torch);
# dist_only with 1024-point tiles
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4k; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 600 python scripts/exp_pairs.py > $OUT/exp_pairs.txt 2>&1; grep -v amdgpu $OUT/exp_pairs.txt | tail -14
timeout -k 5 300 python bench.py --steps 20 --no-cpu-baseline --workload dist_only > $OUT/bench_dist_only.json 2> $OUT/bench_dist_only.err; python - $OUT/bench_dist_only.json <<'PY'
import json,sys
t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
print("dist_only: step %.3f kernel %.3f pts/s %.3e verified %s tile %s" % (d["ms_per_step"], r["kernel_ms_avg"], d["value"], d.get("verified"), d["config"].get("tile_points")))
PY
