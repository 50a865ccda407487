#!/bin/bash
# round 3, session 38: C2-dense / C3-dense with the feature texels at padded strides (producer-side layout; the L2 sees only ~2.5 MiB of 1536-byte-stride data)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_l2probe; mkdir -p $OUT
timeout -k 5 200 python scripts/exp_texel_stride.py c2_dense > $OUT/texel_stride_c2_dense.txt 2>&1; grep -v amdgpu $OUT/texel_stride_c2_dense.txt | cut -c1-170
