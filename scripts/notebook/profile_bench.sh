#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of one bench.py workload.
#   scripts/profile_bench.sh <workload> <tag>
# Summaries land in gpurun_out/prof_<tag>/ ; copy the ones to be judged into profiles/.
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip traces).
set -u
WL=${1:-c2_dense}; TAG=${2:-$WL}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline"
cd /tmp
timeout -k 5 100 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $PMC | tr ' ' '_')
  timeout -k 5 100 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc_$N -o pmc --output-format csv -- $CMD > $OUT/bench_pmc_$N.json 2> $OUT/pmc_$N.err
done
cd $REPO
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
