#!/bin/bash
# Round 6: SQ / LDS / TCP counter passes of the fused kernels, per workload -> gpurun_out/<tag>/<name>_sq.txt (copied to profiles/<tag>/).
# Busy fractions are computed against the MEASURED clock: cycles = GRBM_GUI_ACTIVE / 8 XCDs of the same pass, never 2.4 GHz.
#   scripts/r6_counters.sh <tag> [workload[:points] ...]          (TCP=1: add the vector-memory passes)
set -u
TAG=${1:-r6_v2}; shift
WLS=${@:-c2_patch c3_patch ref_patch c4_patch c2_patch:random c4_patch:random ref_patch:surface c2_patch_f16 c2_dense c3_dense c2_dense_f16 dist_only c5_track}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
FP=$(cd $REPO && python -c "from d3fields_amd import build; print(build.source_fingerprint())")
cd /tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
)
if [ "${TCP:-0}" = 1 ]; then
PASSES+=(
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
 "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"
)
fi
for SPEC in $WLS; do
  WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
  NAME=$WL; [ "$PTS" != grid ] && NAME=${WL}_$PTS
  CMD="python $REPO/bench.py --workload $WL --points $PTS --steps 6 --warmup 2 --no-cpu-baseline --no-verify --traffic off"
  i=0
  for PMC in "${PASSES[@]}"; do
    i=$((i+1))
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/$NAME/p$i -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/$NAME.p$i.err
  done
  (cd $REPO; echo "rocprofv3 --kernel-trace --pmc passes of: $CMD"; echo "source_fingerprint: $FP"; python scripts/summarize_sq.py $OUT/$NAME) > $OUT/${NAME}_sq.txt 2>&1
  rm -rf $OUT/$NAME
done
cd $REPO
ls $OUT/*_sq.txt
