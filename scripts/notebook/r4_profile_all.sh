#!/bin/bash
# Round 4: rocprofv3 evidence for every single-GPU workload: kernel-trace stats + separate PMC passes (FETCH_SIZE,
# WRITE_SIZE, TCC hit/miss, EA read requests / latency), summaries under gpurun_out/<tag>/ -> copied to profiles/.
#   scripts/r3_profile_all.sh <tag> [workloads...]
set -u
TAG=${1:-r4_v1}; shift
WLS=${@:-c2_dense c3_dense c2_patch c3_patch c4_patch c4_dense c5_track ref_patch dist_only}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for WL in $WLS; do
  CMD="python $REPO/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-verify"
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/$WL/trace -o trace --output-format csv -- $CMD > $OUT/$WL.bench_trace.json 2> $OUT/$WL.trace.err
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
    N=$(echo $PMC | tr ' ' '_')
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/$WL/pmc_$N -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/$WL.pmc_$N.err
  done
  (cd $REPO; echo "rocprofv3 --kernel-trace --stats / --pmc passes of: $CMD"; python scripts/summarize_prof.py $OUT/$WL) > $OUT/${WL}_summary.txt 2>&1
  rm -rf $OUT/$WL/trace/*/*hip_api* 2>/dev/null
done
cd $REPO
ls $OUT/*_summary.txt
