#!/bin/bash
# round 5, session 32 (PRODUCT build): the multi-view instance association (align_instance_mask_v3 and its stages) against the reference's goldens and the oracle
set -u
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_callers.py -m gpu -q -x -k "align or compose" 2>&1 | grep -v amdgpu | tail -25 | cut -c1-400
