#!/bin/bash
# round 6, session 4 (PRODUCT build): the pairwise MFMA kernel with two stage buffers (kernel-trace of c5), the corr tests, the combined
# probe test, then the whole GPU suite on the sources as they stand (window kernel back on the pipelined VALU loop)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s4; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_walks.py -x -q -m gpu -k "pairwise or similarity or corr or probe or golden" 2>&1 | tail -3 | cut -c1-200
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/c5trace -o trace --output-format csv -- python $REPO/bench.py --workload c5_track --steps 20 --warmup 2 --no-cpu-baseline > $OUT/c5_trace.json 2> $OUT/c5_trace.err
(cd $REPO; python scripts/summarize_prof.py $OUT/c5trace) > $OUT/c5_track_trace.txt 2>&1; rm -rf $OUT/c5trace
head -8 $OUT/c5_track_trace.txt | cut -c1-200
cd $REPO
python - $OUT/c5_trace.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print("c5 step %.4f ms value %.4g verified %s" % (d["ms_per_step"], d["value"], d["verified"]))
PY
timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
