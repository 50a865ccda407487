#!/bin/bash
# round 3, session 19: masked instance clouds, voxel-grid mean, the one-launch tracking step after the load reordering
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3u; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_callers.py -m gpu -q -x -k "masked or voxel or aggr or pcd or rigid or tracking or driver" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
timeout -k 5 600 python scripts/exp_callers.py rigid > $OUT/rigid.txt 2>&1; grep -v amdgpu.ids $OUT/rigid.txt | tail -5
