#!/bin/bash
# round 6, session 9 (PRODUCT build): where do the dense kernel's L2 fills come from -- HBM or the Infinity Cache?
# TCC_EA0_RDREQ (all fabric read requests) against TCC_EA0_RDREQ_DRAM (those that go to DRAM), the same for writes; C2-dense, C3-dense, C4-dense
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s9; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for WL in c2_dense c3_dense c2_patch; do
  for PMC in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum GRBM_GUI_ACTIVE"; do
    N=$(echo $PMC | cut -c1-30 | tr ' ' '_')
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/$WL/$N -o pmc --output-format csv -- python $REPO/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-verify --traffic off > /dev/null 2> $OUT/${WL}_$N.err
  done
  echo "== $WL"; (cd $REPO; python scripts/summarize_prof.py $OUT/$WL) 2>&1 | grep -E "fused_eval" | grep "TCC\|GRBM" | cut -c1-130 | tee $OUT/${WL}_ea.txt
  rm -rf $OUT/$WL
done
