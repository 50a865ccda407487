#!/bin/bash
# round 6, session 30 (PRODUCT build, final sources): soak -- the GPU suite twice more back to back, 80 more seeds of the large-cloud fuzz,
# the register-rows tests twenty times over (index mode, counted waits and the cell ring under repetition), the seeded fuzz shapes with 1024 channels
set -u
export TMPDIR=/tmp
for i in 1 2; do timeout -k 5 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -1 | cut -c1-160; done
timeout -k 5 900 python scripts/notebook/exp_fuzz_clouds.py 200 80 2>&1 | grep -v amdgpu | grep -v "^ok" | tail -5
for i in $(seq 1 20); do timeout -k 5 300 python -m pytest tests/test_gpu_walks.py -q -x -m gpu -k "register_rows or point_order or cloud_gate" 2>&1 | tail -1 | cut -c1-80; done | sort | uniq -c
