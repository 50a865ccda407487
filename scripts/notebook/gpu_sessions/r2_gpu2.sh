#!/bin/bash
# Round 2, GPU call 2: full GPU suite (no -x), probe timing, SQ counters of the cell-run gather vs the direct gather.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2b; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline"
timeout -k 5 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -15 $OUT/pytest.log
timeout -k 5 200 $B --workload c2_dense --steps 20 > $OUT/bench_c2_dense.json 2> $OUT/bench_c2_dense.err
cd /tmp
for VAR in direct runs; do
  if [ $VAR = direct ]; then export D3F_EXP_RUNS=-1; else export D3F_EXP_RUNS=8 D3F_EXP_RUNS_OCC=5; fi
  CMD="$B --workload c2_patch --steps 5 --warmup 1 --no-verify"
  i=0
  for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc_$VAR/p$i -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_${VAR}_p$i.err
  done
done
unset D3F_EXP_RUNS D3F_EXP_RUNS_OCC
cd $REPO
python - "$OUT" <<'PY' > $OUT/pmc_c2_patch_summary.txt 2>&1
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for var in ("direct", "runs"):
    agg = defaultdict(lambda: defaultdict(list))
    for p in glob.glob(os.path.join(root, "pmc_" + var, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "fused_eval" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(var, k)
        for c, v in sorted(cs.items()):
            print("   %-34s %18.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUT/pmc_c2_patch_summary.txt
grep -h '^{' $OUT/bench_c2_dense.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('c2_dense step %.3f kernel %.3f stepdev %.3f value %.3e verified %s order %s' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['step_device_ms_avg'], d['value'], d['verified'], d['config'].get('point_order')))"
