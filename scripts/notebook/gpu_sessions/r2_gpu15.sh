#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2o; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "cell_run" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>&1; }
for WL in c2_patch c3_patch c4_patch; do
  run base $WL D3F_EXP_RUNS_MOVES=0
  run moves $WL D3F_EXP_RUNS_MOVES=1
  run moves_u1k8 $WL D3F_EXP_RUNS_MOVES=1 D3F_EXP_RUNS_U=1 D3F_EXP_RUNS=8
done
run moves_random c2_patch D3F_EXP_RUNS_MOVES=1
cd /tmp
for MV in 0 1; do
D3F_EXP_RUNS_MOVES=$MV timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_mv$MV -o pmc --output-format csv -- $B --workload c4_patch --steps 5 --warmup 1 --no-verify > /dev/null 2> $OUT/pmc_mv$MV.err
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for var in ("pmc_mv0", "pmc_mv1"):
    agg = defaultdict(lambda: defaultdict(list))
    for p in glob.glob(os.path.join(root, var, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "fused_eval" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(var, k, {c: int(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms (min %.3f) | frac %.3f | verified %s | %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
