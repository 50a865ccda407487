#!/bin/bash
# evidence for the new c2_dense default (channel-sliced, 16 points per workgroup)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2_final5; mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/r2_profile_all.sh r2_v3 c2_dense > $OUT/profile.log 2>&1; tail -1 $OUT/profile.log
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r2_v3/default_bench -o trace --output-format csv -- python $REPO/bench.py > $REPO/gpurun_out/r2_v3/default_bench_under_rocprof.json 2> $OUT/default_bench.err
for PMC in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  N=$(echo $PMC | tr ' ' '_')
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $PMC -d $REPO/gpurun_out/r2_v3/c2_dense/pmc2_$N -o pmc --output-format csv -- python $REPO/bench.py --workload c2_dense --steps 10 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc2_$N.err
done
cd $REPO
python scripts/summarize_prof.py gpurun_out/r2_v3/default_bench > gpurun_out/r2_v3/default_bench_kernel_stats.txt 2>&1
python scripts/summarize_prof.py gpurun_out/r2_v3/c2_dense > gpurun_out/r2_v3/c2_dense_summary_full.txt 2>&1
tail -1 gpurun_out/r2_v3/default_bench_under_rocprof.json | cut -c1-200
grep -i "sliced" gpurun_out/r2_v3/c2_dense_summary_full.txt | tail -20
rm -rf gpurun_out/r2_v3/*/trace/*/*hip_api* gpurun_out/r2_v3/default_bench/*/*hip_api* 2>/dev/null
