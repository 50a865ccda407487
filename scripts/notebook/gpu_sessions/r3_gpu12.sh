#!/bin/bash
# round 3, session 12: config 4 on its lattice slab (window kernel / brick walk), bench --gpus 2 as ONE command (gloo ranks
# on one GPU), the new caller tests
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3n; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_callers.py -m gpu -q -x -k "bench_workload or sc1 or masked_pixel or fps_pixels or select_features_rand_v2 or driver_sequence" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for WL in c4_patch c4_dense; do for P in grid random; do
  timeout -k 5 300 python bench.py --workload $WL --points $P --steps 10 --no-cpu-baseline > $OUT/bench_${WL}_$P.json 2> $OUT/bench_${WL}_$P.err
done; done
timeout -k 5 300 python bench.py --gpus 2 --backend gloo --steps 5 --no-cpu-baseline --workload c2_patch > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "2-rank rc=$?"
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r3n/bench_*.json")):
    try:
        d = json.loads([l for l in open(p) if l.startswith("{")][-1]); r = d["roofline"]
        print("%-34s step %.3f kernel %.3f frac %.3f verified %s n_gpus %d %s | %s" % (p.split("/")[-1], d["ms_per_step"], r["kernel_ms_avg"], r["frac"], d["verified"], d["n_gpus"], r["kernel"], d["config"].get("rank_devices")))
    except Exception as e:
        print(p, "ERR", e, open(p.replace(".json", ".err")).read()[-400:])
PY
