#!/bin/bash
# round 3, session 18: the optimiser step of rigid_tracking as ONE launch (d3f_track_step)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3t; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_callers.py -m gpu -q -x -k "rigid or tracking or driver_sequence" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
timeout -k 5 600 python scripts/exp_callers.py rigid > $OUT/rigid.txt 2>&1; grep -v amdgpu.ids $OUT/rigid.txt
cd /tmp && timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_rigid -o rigid --output-format csv -- python $REPO/scripts/exp_callers.py rigid > /dev/null 2> $OUT/prof_rigid.err; cd $REPO
python scripts/kernel_stats.py $OUT/prof_rigid d3f:: > $OUT/rigid_kernel_stats.txt; head -12 $OUT/rigid_kernel_stats.txt
