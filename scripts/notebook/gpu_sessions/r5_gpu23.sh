#!/bin/bash
# round 5, session 23 (PRODUCT build): the new exact-Hilbert-order test; then the whole GPU suite
set -u
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "hilbert_order" 2>&1 | grep -v amdgpu | tail -12 | cut -c1-300
timeout -k 5 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300
