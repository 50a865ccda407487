#!/bin/bash
# round 6, session 13 (PRODUCT build): the (2, 8) cell-run kernel with TWO remembered cells per view (195 VGPRs, two waves per SIMD) --
# bit-identity tests, then C4-patch cloud / ref_patch surface / a 200 k cloud on 1024-d maps against round 6's numbers (2.67-2.71 / 0.135-0.138 ms)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s13; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py tests/test_gpu_fuzz.py -x -q -m gpu -k "cell_run or cloud or fuzz or seeded_eval or bench_workload" 2>&1 | tail -3 | cut -c1-200
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-34s value %.4g step %.4f ms kernel avg %.4f min %.4f frac %.3f verified %s %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"][:40]))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
for i in 1 2; do
for SPEC in c4_patch:random ref_patch:surface; do
  WL=${SPEC%%:*}; PTS=${SPEC##*:}
  timeout -k 5 300 python bench.py --no-cpu-baseline --traffic off --steps 30 --workload $WL --points $PTS > $OUT/${WL}_$PTS.json 2> $OUT/${WL}_$PTS.err
  line $OUT/${WL}_$PTS.json "$WL $PTS"
done
done
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $OUT/pmc -o pmc --output-format csv -- python $REPO/bench.py --workload c4_patch --points random --steps 6 --warmup 2 --no-cpu-baseline --no-verify --traffic off > /dev/null 2> $OUT/pmc.err
(cd $REPO; python scripts/summarize_sq.py $OUT/pmc) 2>&1 | grep -E "^==|TCP_TOTAL|per cycle per CU|derived" | head -8 | cut -c1-160
