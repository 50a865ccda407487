#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r2j; mkdir -p $OUT
export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --steps 20"
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "sliced or lattice_walk" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -12 $OUT/pytest.log
run() { TAG=$1; WL=$2; shift 2; env "$@" timeout -k 5 300 $B --workload $WL > $OUT/bench_${WL}_$TAG.json 2>&1; }
for WL in c2_dense c3_dense; do
  run base $WL D3F_EXP_SLICED=0
  run sl128 $WL D3F_EXP_SLICED=1
  run sl256 $WL D3F_EXP_SLICED=2
done
cd /tmp
for VAR in 1 2; do
  for PMC in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    N=$(echo $PMC | tr ' ' '_')
    D3F_EXP_SLICED=$VAR timeout -k 5 120 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc_sl$VAR/$N -o pmc --output-format csv -- $B --workload c2_dense --steps 5 --warmup 1 --no-verify > /dev/null 2> $OUT/pmc_sl${VAR}_$N.err
  done
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for var in ("pmc_sl1", "pmc_sl2"):
    agg = defaultdict(lambda: defaultdict(list))
    for p in glob.glob(os.path.join(root, var, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "fused_eval" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(var, k)
        for c, v in sorted(cs.items()):
            print("   %-34s %18.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
for f in $OUT/bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t)
    r=d["roofline"]
    print("step %.3f ms | kernel %.3f ms (min %.3f) | frac %.3f | verified %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified")))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
)"; done
