#!/bin/bash
# round 5, session 31 (PRODUCT build): the new large-cloud fuzz cases of the test-suite, then a longer sweep of the same generator (60 more seeds)
set -u
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "large_cloud" 2>&1 | grep -v amdgpu | tail -8 | cut -c1-400
timeout -k 5 1500 python scripts/notebook/exp_fuzz_clouds.py 100 60 2>&1 | grep -v amdgpu | tail -70 | cut -c1-260
