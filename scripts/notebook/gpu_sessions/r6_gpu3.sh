#!/bin/bash
# round 6, session 3 (EXPERIMENTS build): the matrix-core point loop with per-k-step skipping and prefetched B reads against the pipelined
# VALU loop (same box), phase stamps of both, matrix-core counters; the LDS probe again (rates from kernel time); kernel-trace stats of c5
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s3; mkdir -p $OUT
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-34s value %.4g step %.4f ms kernel avg %.4f min %.4f frac %.3f verified %s  %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"][:70]))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
(cd /tmp && hipcc --offload-arch=gfx950 -O3 $REPO/scripts/notebook/microbench/lds_read_rate.hip -o /tmp/lds_read_rate 2>/dev/null && timeout -k 5 300 /tmp/lds_read_rate) > $OUT/lds_read_rate.txt 2>&1
export D3F_BUILD_EXPERIMENTS=1
python -m d3fields_amd.build > $OUT/build_exp.log 2>&1
for MM in -1 0 -1 0; do
  for SPEC in c2_patch ref_patch c4_patch; do
    WL=${SPEC%%:*}; PTS=grid
    D3F_EXP_WINDOW_MFMA=$MM timeout -k 5 300 python bench.py --no-cpu-baseline --no-verify --steps 30 --workload $WL --points $PTS > $OUT/exp${MM}_${WL}_$PTS.json 2> $OUT/exp${MM}_${WL}_$PTS.err
    line $OUT/exp${MM}_${WL}_$PTS.json "exp mfma=$MM $WL $PTS"
  done
done
for MM in -1 0; do
  echo "== phase stamps, D3F_EXP_WINDOW_MFMA=$MM"
  D3F_EXP_WINDOW_MFMA=$MM D3F_EXP_STAMPS=1 timeout -k 5 300 python scripts/notebook/exp_stamps.py c2_patch c4_patch ref_patch 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_mfma$MM.txt | cut -c1-120
done
cd /tmp
for WL in c2_patch c4_patch; do
  for PMC in "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    N=$(echo $PMC | cut -c1-20 | tr ' ' '_')
    D3F_EXP_WINDOW_MFMA=0 timeout -k 5 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc_${WL}/$N -o pmc --output-format csv -- python $REPO/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_${WL}_$N.err
  done
  (cd $REPO; python scripts/summarize_sq.py $OUT/pmc_${WL}) > $OUT/${WL}_mfma_sq.txt 2>&1
  rm -rf $OUT/pmc_${WL}
done
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/c5trace -o trace --output-format csv -- python $REPO/bench.py --workload c5_track --steps 10 --warmup 2 --no-cpu-baseline --no-verify > $OUT/c5_trace.json 2> $OUT/c5_trace.err
(cd $REPO; python scripts/summarize_prof.py $OUT/c5trace) > $OUT/c5_track_trace.txt 2>&1; rm -rf $OUT/c5trace
head -16 $OUT/c5_track_trace.txt | cut -c1-200
cd $REPO
grep -E "^==|MFMA|busy|parked|per SIMD" $OUT/c2_patch_mfma_sq.txt | head -30
