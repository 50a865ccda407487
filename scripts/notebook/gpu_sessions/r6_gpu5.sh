#!/bin/bash
# round 6, session 5 (EXPERIMENTS build): the channel-sliced kernel with SLICE-PINNED XCDs (D3F_EXP_SLICED_PIN=1: units in slice-major
# order, a contiguous eighth per XCD) against the round-robin unit mapping of rounds 2-5, same box; fabric reads of both (FETCH_SIZE)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s5; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d["roofline"]
    print("%-40s value %.4g step %.4f ms kernel avg %.4f min %.4f frac %.3f verified %s  %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], d.get("verified"), r["kernel"][:60]))
except Exception as e:
    print(sys.argv[2], "ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
python -m d3fields_amd.build > $OUT/build_exp.log 2>&1
for PIN in 0 1 0 1; do
  for SPEC in c2_dense c3_dense c2_dense_f16 c2_dense:random; do
    WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
    D3F_EXP_SLICED_PIN=$PIN timeout -k 5 300 python bench.py --no-cpu-baseline --steps 30 --workload $WL --points $PTS > $OUT/pin${PIN}_${WL}_$PTS.json 2> $OUT/pin${PIN}_${WL}_$PTS.err
    line $OUT/pin${PIN}_${WL}_$PTS.json "pin=$PIN $WL $PTS"
  done
done
for UNIT in 64 32; do
  D3F_EXP_SLICED_PIN=1 D3F_EXP_SLICED_UNIT=$UNIT timeout -k 5 300 python bench.py --no-cpu-baseline --no-verify --steps 30 --workload c2_dense > $OUT/pin1_unit${UNIT}.json 2> $OUT/pin1_unit${UNIT}.err
  line $OUT/pin1_unit${UNIT}.json "pin=1 unit=$UNIT c2_dense"
done
cd /tmp
for PIN in 0 1; do
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $PMC | tr ' ' '_')
    D3F_EXP_SLICED_PIN=$PIN timeout -k 5 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc_pin$PIN/$N -o pmc --output-format csv -- python $REPO/bench.py --workload c2_dense --steps 6 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_pin${PIN}_$N.err
  done
  (cd $REPO; python scripts/summarize_prof.py $OUT/pmc_pin$PIN) 2>&1 | grep -E "sliced" | cut -c1-150 | tee $OUT/pin${PIN}_c2_dense_pmc.txt
  rm -rf $OUT/pmc_pin$PIN
done
