#!/bin/bash
# round 6, session 59 (PRODUCT build, FINAL sources): soak -- the GPU suite twice more back to back, 60 more seeds of the large-cloud fuzz (phase A of every
# fused kernel was recompiled by the d3f_device.h refactor), the distance / grid tests ten times over
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r6_s59
{
for i in 1 2; do timeout -k 5 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -1 | cut -c1-160; done
timeout -k 5 900 python scripts/notebook/exp_fuzz_clouds.py 300 60 2>&1 | grep -v amdgpu | grep -v "^ok" | tail -5
for i in $(seq 1 10); do timeout -k 5 300 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -q -x -m gpu -k "dist or grid or shell" 2>&1 | tail -1 | cut -c1-80; done | sort | uniq -c
} 2>&1 | tee gpurun_out/r6_s59/soak.txt
