#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_phase; mkdir -p $OUT
timeout -k 5 200 python scripts/exp_phase_timing.py c2_dense > $OUT/phase_c2_dense.txt 2>&1; grep -v amdgpu $OUT/phase_c2_dense.txt
timeout -k 5 200 python scripts/exp_phase_timing.py c4_dense > $OUT/phase_c4_dense.txt 2>&1; grep -v amdgpu $OUT/phase_c4_dense.txt
