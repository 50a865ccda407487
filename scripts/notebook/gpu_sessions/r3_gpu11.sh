#!/bin/bash
# round 3, session 11: the whole GPU suite on the product build after the clean-up (fuse_common.h split, staged kernel and
# environment knobs gone, D3F_TUNE_DIRECT_GATHER as the tests' reference), then the experiments build's extra sweeps
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3m; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_product.log 2>&1; echo "product rc=$?"; tail -4 $OUT/pytest_product.log
python bench.py --steps 20 --no-cpu-baseline > $OUT/bench_c2_dense.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3m/bench_c2_dense.json") if l.startswith("{")][-1]); r=d["roofline"]
print("c2_dense step %.3f kernel %.3f frac %.3f verified %s %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["frac"], d["verified"], r["kernel"]))
PY
export D3F_BUILD_EXPERIMENTS=1
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_abi.py -m "gpu or not gpu" -q -x > $OUT/pytest_exp.log 2>&1; echo "experiments rc=$?"; tail -4 $OUT/pytest_exp.log
