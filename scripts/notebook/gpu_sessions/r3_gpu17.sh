#!/bin/bash
# round 3, session 17: multi-workgroup FPS (tests + timing), caller-side kernel timings and their rocprofv3 kernel traces
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3s; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_callers.py -m gpu -q -x -k "fps or select_features or masked_pixel or driver_sequence or rigid" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout -k 5 600 python scripts/exp_callers.py > $OUT/callers.txt 2>&1; grep -v amdgpu.ids $OUT/callers.txt
cd /tmp && timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_callers -o callers --output-format csv -- python $REPO/scripts/exp_callers.py > /dev/null 2> $OUT/prof_callers.err; cd $REPO
python scripts/kernel_stats.py $OUT/prof_callers d3f:: > $OUT/callers_kernel_stats.txt; head -40 $OUT/callers_kernel_stats.txt
