#!/bin/bash
# round 5, session 16 (PRODUCT build): the new map_check_many / word-ring test, smoke(), the default bench line
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_s16
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "map_check" 2>&1 | grep -v amdgpu | tail -12 | cut -c1-300


