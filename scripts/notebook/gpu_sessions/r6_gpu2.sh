#!/bin/bash
# round 6, session 2 (PRODUCT build, then an EXPERIMENTS build for the A/B): first run of the matrix-core point loop of the window
# kernel (bit-identity tests, then times against the pipelined VALU loop on the same box), the guarded MFMA pairwise kernel, the
# small-cloud changes (local caller order, one probe launch, five ordering launches), the finite-word ring test, the 8-rank dry run
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_s2; mkdir -p $OUT
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
timeout -k 5 900 python -m pytest tests/test_gpu_walks.py -x -q -m gpu -k "window or lattice_walk or cloud_gate or map_order or fp16_stored" 2>&1 | tail -4 | cut -c1-200
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pairwise or ring or golden or similarity or corr" 2>&1 | tail -4 | cut -c1-200
timeout -k 5 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "pairwise" 2>&1 | tail -3 | cut -c1-200
for SPEC in c2_patch c3_patch ref_patch c4_patch c2_patch:random c3_patch:random ref_patch:random ref_patch:surface c5_track c2_dense; do
  WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
  timeout -k 5 300 python bench.py --no-cpu-baseline --steps 30 --workload $WL --points $PTS > $OUT/prod_${WL}_$PTS.json 2> $OUT/prod_${WL}_$PTS.err
  line $OUT/prod_${WL}_$PTS.json "prod $WL $PTS"
done
# the same box, experiments build: the pipelined VALU loop of rounds 4-5 against the matrix-core loop
export D3F_BUILD_EXPERIMENTS=1
python -m d3fields_amd.build > $OUT/build_exp.log 2>&1
for MM in -1 0; do
  for SPEC in c2_patch c3_patch ref_patch c4_patch c2_patch:random; do
    WL=${SPEC%%:*}; PTS=grid; [ "$SPEC" != "$WL" ] && PTS=${SPEC##*:}
    D3F_EXP_WINDOW_MFMA=$MM timeout -k 5 300 python bench.py --no-cpu-baseline --no-verify --steps 30 --workload $WL --points $PTS > $OUT/exp${MM}_${WL}_$PTS.json 2> $OUT/exp${MM}_${WL}_$PTS.err
    line $OUT/exp${MM}_${WL}_$PTS.json "exp mfma=$MM $WL $PTS"
  done
done
# surface cloud: caller order vs forced Hilbert order (D3F_TUNE_FORCE_REORDER = 1 << 14)
timeout -k 5 300 python bench.py --no-cpu-baseline --no-verify --steps 30 --workload ref_patch --points surface --tuning 0x4000 > $OUT/exp_surface_forced.json 2> $OUT/exp_surface_forced.err
line $OUT/exp_surface_forced.json "surface, forced Hilbert order"
unset D3F_BUILD_EXPERIMENTS
python -m d3fields_amd.build --force > $OUT/build_prod.log 2>&1
timeout -k 5 900 python -m pytest tests/test_gpu_sharding.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | cut -c1-300
