#!/bin/bash
# round 5, session 33 (PRODUCT build, final sources): the driver's sequence -- GPU suite with -x, smoke(), default bench -- then the multi-rank
# bench plumbing on ONE GPU once more (--force-dist over RCCL with one rank; two gloo ranks sharing the GPU; the default workload with two gloo ranks)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_s33
timeout -k 5 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -2 | cut -c1-200
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 5 600 python bench.py > gpurun_out/r5_s33/default.json 2> gpurun_out/r5_s33/default.err
timeout -k 5 600 python bench.py --gpus 1 --force-dist --steps 10 --no-cpu-baseline > gpurun_out/r5_s33/c2_dense_force_dist.json 2> gpurun_out/r5_s33/force_dist.err
timeout -k 5 600 python bench.py --gpus 2 --backend gloo --steps 10 --workload c2_patch --no-cpu-baseline > gpurun_out/r5_s33/c2_patch_gloo_2ranks.json 2> gpurun_out/r5_s33/gloo.err
timeout -k 5 600 python bench.py --gpus 2 --backend gloo --steps 10 --no-cpu-baseline > gpurun_out/r5_s33/c2_dense_gloo_2ranks.json 2> gpurun_out/r5_s33/gloo2.err
python - <<'PY'
import json
for f in ("default", "c2_dense_force_dist", "c2_patch_gloo_2ranks", "c2_dense_gloo_2ranks"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5_s33/%s.json" % f) if l.startswith("{")][-1])
        print(f, {k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "single_rank_points_per_s_same_workload", "value_full_field", "scaling_efficiency", "verified")},
              "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d["config"]["workload"][:30])
    except Exception as e:
        print(f, "ERR", e)
PY
