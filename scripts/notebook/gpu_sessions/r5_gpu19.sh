#!/bin/bash
# round 5, session 19 (PRODUCT build): the whole GPU suite (stall sentinel of d3f_track_run, map_check_many, everything)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_s19
timeout -k 5 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tee gpurun_out/r5_s19/pytest.txt | tail -8 | cut -c1-300
