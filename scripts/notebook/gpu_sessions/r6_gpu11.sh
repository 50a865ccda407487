#!/bin/bash
# round 6, session 11 (PRODUCT build): pairwise_mfma_kernel held to four waves per SIMD (__launch_bounds__(256, 4): its 16 accumulator
# registers no longer cost a wave); kernel-trace of c5, corr tests
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s11; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "pairwise or similarity or corr" 2>&1 | tail -2 | cut -c1-200
cd /tmp
for i in 1 2; do
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/c5trace -o trace --output-format csv -- python $REPO/bench.py --workload c5_track --steps 20 --warmup 2 --no-cpu-baseline --traffic off > $OUT/c5_trace.json 2> $OUT/c5_trace.err
(cd $REPO; python scripts/summarize_prof.py $OUT/c5trace) > $OUT/c5_track_trace.txt 2>&1; rm -rf $OUT/c5trace
sed -n 4,6p $OUT/c5_track_trace.txt | cut -c1-170
done
cd $REPO
timeout -k 5 300 python bench.py --workload c5_track --steps 30 --no-cpu-baseline --traffic off | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c5 step %.4f ms value %.4g verified %s' % (d['ms_per_step'], d['value'], d['verified']))"
