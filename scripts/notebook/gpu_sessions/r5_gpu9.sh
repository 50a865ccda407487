#!/bin/bash
# round 5, session 9 (PRODUCT build): the whole GPU suite with the fp16 window / sliced kernels
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_s9
timeout -k 5 2400 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s9/pytest.txt | tail -30 | cut -c1-300
