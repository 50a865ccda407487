#!/bin/bash
# Round 5: rocprofv3 evidence for every single-GPU workload: kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# TCC hit/miss, EA read requests / latency, SQ instruction counts), one summary per workload and point set under
# gpurun_out/<tag>/ -> copied to profiles/<tag>/.  Every summary is stamped with the source fingerprint of the library it profiled
# (d3fields_amd/build.py), which scripts/make_traffic_json.py carries into profiles/traffic.json.
#   scripts/r5_profile_all.sh <tag> [workload[:points] ...]
set -u
TAG=${1:-r5_v1}; shift
WLS=${@:-c2_dense c3_dense c2_patch c3_patch c4_patch ref_patch c2_patch:random c3_patch:random ref_patch:random ref_patch:surface c4_patch:random c5_track c2_patch_f16 c2_dense_f16 dist_only}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
FP=$(cd $REPO && python -c "from d3fields_amd import build; print(build.source_fingerprint())")
cd /tmp
for SPEC in $WLS; do
  WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
  NAME=$WL; [ "$PTS" != grid ] && NAME=${WL}_$PTS
  CMD="python $REPO/bench.py --workload $WL --points $PTS --steps 10 --warmup 2 --no-cpu-baseline --no-verify"
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/$NAME/trace -o trace --output-format csv -- $CMD > $OUT/$NAME.bench_trace.json 2> $OUT/$NAME.trace.err
  if [ "${NO_PMC:-0}" != 1 ]; then
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
    N=$(echo $PMC | tr ' ' '_')
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/$NAME/pmc_$N -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/$NAME.pmc_$N.err
  done
  fi
  (cd $REPO; echo "rocprofv3 --kernel-trace --stats / --pmc passes of: $CMD"; echo "source_fingerprint: $FP"; python scripts/summarize_prof.py $OUT/$NAME) > $OUT/${NAME}_summary.txt 2>&1
  rm -rf $OUT/$NAME/trace/*/*hip_api* $OUT/$NAME 2>/dev/null
done
cd $REPO
ls $OUT/*_summary.txt
