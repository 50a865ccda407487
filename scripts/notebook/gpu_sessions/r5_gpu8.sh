#!/bin/bash
# round 5, session 8 (PRODUCT build): fp16-stored maps on the window / sliced kernels -- parity tests, then the f16 bench workloads beside the fp32 ones
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_s8
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -m gpu -q -x -k "fp16 or float16 or f16 or window" 2>&1 | grep -v amdgpu | tail -15 | cut -c1-300
for WL in c2_patch_f16 c2_patch c2_dense_f16 c2_dense c3_patch ref_patch c4_patch; do
  timeout -k 5 300 python bench.py --workload $WL --no-cpu-baseline --steps 30 > gpurun_out/r5_s8/$WL.json 2> gpurun_out/r5_s8/$WL.err
  python - gpurun_out/r5_s8/$WL.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-14s step %.4f kernel avg %.4f min %.4f frac %.3f verified %s %s" % (sys.argv[1].split('/')[-1][:-5], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
