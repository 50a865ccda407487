#!/bin/bash
# round 5, session 10 (PRODUCT build): single-launch scan in the ordering, batched map checks -- GPU suite, then the c5 step
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_s10
timeout -k 5 2400 python -m pytest tests/test_gpu_walks.py tests/test_gpu_callers.py -m gpu -q -x 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s10/pytest.txt | tail -8 | cut -c1-300
for WL in c5_track; do
  for i in 1 2; do
  timeout -k 5 300 python bench.py --workload $WL --no-cpu-baseline --steps 50 > gpurun_out/r5_s10/$WL.json 2> gpurun_out/r5_s10/$WL.err
  python - gpurun_out/r5_s10/$WL.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-14s step %.4f kernel avg %.4f min %.4f step_device %.4f static-maps pts/s %.4g verified %s %s" % (sys.argv[1].split('/')[-1][:-5], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["step_device_ms_avg"], d.get("points_per_s_with_static_maps", 0), d.get("verified"), r["kernel"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  done
done
REPO=$(pwd); cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r5_s10/c5/trace -o trace --output-format csv -- python $REPO/bench.py --workload c5_track --no-cpu-baseline --no-verify --steps 30 > /dev/null 2> $REPO/gpurun_out/r5_s10/c5.trace.err
cd $REPO; python scripts/summarize_prof.py gpurun_out/r5_s10/c5 > gpurun_out/r5_s10/c5_track_summary.txt 2>&1; rm -rf gpurun_out/r5_s10/c5/trace/*/*hip_api*; head -32 gpurun_out/r5_s10/c5_track_summary.txt | cut -c1-200
