#!/bin/bash
# round 5, session 14 (EXPERIMENTS build): the GPU suite with every knob variant (the 2 tests the product build skips), then the
# multi-rank bench plumbing on ONE GPU: --force-dist over RCCL with one rank, and two gloo ranks sharing the GPU
set -u
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out/r5_s14
timeout -k 5 2400 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v amdgpu | tail -6 | cut -c1-300
unset D3F_BUILD_EXPERIMENTS
python -m d3fields_amd.build > /dev/null 2>&1
timeout -k 5 600 python bench.py --gpus 1 --force-dist --steps 10 --no-cpu-baseline > gpurun_out/r5_s14/force_dist_1rank_rccl.json 2> gpurun_out/r5_s14/force_dist.err; tail -c 1500 gpurun_out/r5_s14/force_dist_1rank_rccl.json | cut -c1-1500; echo
timeout -k 5 600 python bench.py --gpus 2 --backend gloo --steps 10 --workload c2_patch > gpurun_out/r5_s14/two_ranks_gloo_one_gpu.json 2> gpurun_out/r5_s14/gloo.err; python - <<'PY'
import json
for f in ("force_dist_1rank_rccl", "two_ranks_gloo_one_gpu"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5_s14/%s.json" % f) if l.startswith("{")][-1])
        print(f, {k: d.get(k) for k in ("value", "n_gpus", "single_rank_points_per_s_same_workload", "compute_only_points_per_s", "value_full_field", "scaling_efficiency", "verified")}, d["config"]["workload"][:40])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r5_s14/gloo.err | cut -c1-300
