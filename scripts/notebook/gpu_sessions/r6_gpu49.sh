#!/bin/bash
# round 6, session 49 (PRODUCT build): the keypoint pre-filter on the distance-only kernel's arithmetic -- tests, grid_shell timing
set -u
REPO=$(pwd); TAG=${TAG:-r6_s49}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py tests/test_gpu_callers.py -q -m gpu -x -k "grid_shell or select or dist_only or surface or shell" 2>&1 | tail -5 | cut -c1-250
timeout -k 5 600 python scripts/notebook/exp_grid_shell_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/grid_shell_time.txt
