#!/bin/bash
# round 6, session 14 (PRODUCT build, final sources): soak -- the GPU suite twice more back to back, 60 more seeds of the large-cloud fuzz,
# the ordering / probe / gate / ring / pairwise tests ten times over
set -u
export TMPDIR=/tmp
for i in 1 2; do timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -1 | cut -c1-160; done
timeout -k 5 900 python scripts/notebook/exp_fuzz_clouds.py 200 60 2>&1 | grep -v amdgpu | grep -v "^ok" | tail -5
for i in $(seq 1 10); do timeout -k 5 400 python -m pytest tests/test_gpu_walks.py tests/test_gpu_parity.py -q -x -m gpu -k "point_order or cloud_gate or probe or ring or pairwise_guarded or map_check" 2>&1 | tail -1 | cut -c1-80; done | sort | uniq -c
