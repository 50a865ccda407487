#!/bin/bash
# round 5, session 34 (what-if builds -DD3F_PAIRWISE_WHATIF=1/2, scripts/notebook/patches/r5_s34_pairwise_whatif.patch): is the LDS return path
# (8 ds_read_b128 per 64 packed instructions: 64 of 256 cycles per SIMD, x 4 SIMDs = the whole LDS) what keeps pairwise_dist_kernel at 88 % VALU issue?
set -u
export TMPDIR=/tmp
for W in 0 1 2 0; do
  cp build_ab/pairwise_$W.so d3fields_amd/libd3fields_hip.so
  echo "whatif $W"; timeout -k 5 300 python scripts/notebook/exp_pairwise.py 2>&1 | grep -v amdgpu | grep " dist " | head -2
done
