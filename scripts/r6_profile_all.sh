#!/bin/bash
# Round 6: rocprofv3 evidence for every single-GPU workload: kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE, TCC hit / miss,
# EA read requests / latency, SQ instruction counts) AND the SQ / LDS busy counters (scripts/summarize_sq.py: fractions against the measured
# clock), one summary per workload and point set under gpurun_out/<tag>/ -> copied to profiles/<tag>/.  Every summary is stamped with the
# source fingerprint of the library it profiled (d3fields_amd/build.py), which scripts/make_traffic_json.py carries into profiles/traffic.json.
#   scripts/r6_profile_all.sh <tag> [workload[:points] ...]
set -u
TAG=${1:-r6_v2}; shift
WLS=${@:-c2_dense c3_dense c2_patch c3_patch c4_patch ref_patch c2_patch:random c3_patch:random ref_patch:random ref_patch:surface c4_patch:random c5_track c2_patch_f16 c2_dense_f16 dist_only}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
FP=$(cd $REPO && python -c "from d3fields_amd import build; print(build.source_fingerprint())")
cd /tmp
for SPEC in $WLS; do
  WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
  NAME=$WL; [ "$PTS" != grid ] && NAME=${WL}_$PTS
  CMD="python $REPO/bench.py --workload $WL --points $PTS --steps 10 --warmup 2 --no-cpu-baseline --no-verify --traffic off"
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/$NAME/trace -o trace --output-format csv -- $CMD > $OUT/$NAME.bench_trace.json 2> $OUT/$NAME.trace.err
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
    N=$(echo $PMC | tr ' ' '_')
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/$NAME/pmc_$N -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/$NAME.pmc_$N.err
  done
  i=0
  for PMC in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/${NAME}_sq/p$i -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/$NAME.sq$i.err
  done
  (cd $REPO; echo "rocprofv3 --kernel-trace --stats / --pmc passes of: $CMD"; echo "source_fingerprint: $FP"; python scripts/summarize_prof.py $OUT/$NAME;
   echo; echo "== SQ / LDS / vector-L1 counters, busy fractions against the measured clock (scripts/summarize_sq.py) =="; python scripts/summarize_sq.py $OUT/${NAME}_sq) > $OUT/${NAME}_summary.txt 2>&1
  rm -rf $OUT/$NAME/trace/*/*hip_api* $OUT/$NAME $OUT/${NAME}_sq 2>/dev/null
done
cd $REPO
ls $OUT/*_summary.txt
