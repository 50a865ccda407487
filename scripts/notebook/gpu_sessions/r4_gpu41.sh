#!/bin/bash
# round 4, session 41 (PRODUCT build): SQ / LDS counter passes of the FINAL window kernel (non-temporal write-through rows) for
# C2-patch, C4-patch and ref_patch -> profiles/r4_v3/*_sq_counters_final.txt
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4_v3; mkdir -p $OUT
for WL in c2_patch c4_patch ref_patch; do
  bash scripts/pmc_kernel.sh final_$WL --workload $WL --no-verify > $OUT/${WL}_sq_counters_final.txt 2>&1
  tail -24 $OUT/${WL}_sq_counters_final.txt
done
