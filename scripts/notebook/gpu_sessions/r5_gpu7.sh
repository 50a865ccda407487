#!/bin/bash
# round 5, session 7 (PRODUCT build): the whole GPU suite on the Hilbert order + gated cloud windows
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_s7
timeout -k 5 2400 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s7/pytest.txt | tail -60 | cut -c1-400
