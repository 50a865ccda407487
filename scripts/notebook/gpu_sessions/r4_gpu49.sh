#!/bin/bash
# round 4, session 49 (PRODUCT build): the seeded shape sweep against the oracle (tests/test_gpu_fuzz.py)
set -u
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | grep -v amdgpu | tail -40 | cut -c1-400
