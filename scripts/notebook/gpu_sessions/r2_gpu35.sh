#!/bin/bash
# end-of-round evidence, second pass (after the window kernel became the lattice default)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2_final2; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash scripts/r2_profile_all.sh r2_v3 c2_patch c3_patch c5_track c4_patch > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
bash scripts/r2_bench_all.sh r2_v3_bench 2>&1 | tail -14
rm -rf gpurun_out/r2_v3/*/trace/*/*hip_api* 2>/dev/null
