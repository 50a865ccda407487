#!/bin/bash
# round 3, session 37 (experiments build): L2 probe with two-slice items; the XCD's workgroups alternating between 2 / 3 units of the sliced launch
set -u
export D3F_BUILD_EXPERIMENTS=1
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_l2probe; mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/microbench/l2_probe.hip -o /tmp/l2_probe 2> /dev/null
timeout -k 5 60 /tmp/l2_probe > $OUT/l2_probe_timing2.txt 2>&1; grep -E "1024 @|512 @ 2048|512 @ 1024|512 @ 3072|512 @ 1536" $OUT/l2_probe_timing2.txt
S="D3F_EXP_SLICED=3,D3F_EXP_SLICED_VC=2"
EXP_REPS=2 timeout -k 5 300 python scripts/exp_knobs.py c2_dense "base:" "ilv2:$S,D3F_EXP_SLICED_ILV=2" "ilv3:$S,D3F_EXP_SLICED_ILV=3" "ilv2u128:$S,D3F_EXP_SLICED_ILV=2,D3F_EXP_SLICED_UNIT=128" "ilv3u96:$S,D3F_EXP_SLICED_ILV=3,D3F_EXP_SLICED_UNIT=96" "ilv4u64:$S,D3F_EXP_SLICED_ILV=4,D3F_EXP_SLICED_UNIT=64" > $OUT/c2_dense_ilv.txt 2>&1
grep -v amdgpu $OUT/c2_dense_ilv.txt | cut -c1-150
