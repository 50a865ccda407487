#!/bin/bash
# round 4, session 9 (PRODUCT build): whole GPU suite with the new tests (map check, tracker fall-back / busy device, new bench
# workloads), the new bench lines, the one-rank RCCL line of c4_patch, kernel traces of refresh-every-step runs
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4i; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log | cut -c1-220
B="python $REPO/bench.py --steps 20"
show() { python - "$1" <<'PY'
import json,sys
try:
    t=[l for l in open(sys.argv[1]) if l.startswith('{')][-1]; d=json.loads(t); r=d["roofline"]
    print("step %.3f ms | kernel %.3f | %.3e pts/s | static-maps %.3e | frac %.3f | valu %s | verified %s | %s" % (d["ms_per_step"], r["kernel_ms_avg"], d["value"], d.get("points_per_s_with_static_maps", 0), r["frac"], (r.get("valu_issue") or {}).get("frac_of_issue_peak"), d.get("verified"), r["kernel"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-400:])
PY
}
for WL in c5_track ref_patch dist_only; do
  timeout -k 5 400 $B --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; echo "$WL: $(show $OUT/bench_$WL.json)"
done
timeout -k 5 400 $B --no-cpu-baseline --workload c2_dense --refresh-maps > $OUT/bench_c2_dense_refresh.json 2> $OUT/bench_c2_dense_refresh.err; echo "c2_dense refresh: $(show $OUT/bench_c2_dense_refresh.json)"
timeout -k 5 400 $B --no-cpu-baseline --gpus 1 --force-dist > $OUT/bench_rccl1_default.json 2> $OUT/bench_rccl1_default.err; echo "force-dist default(c2_dense at 1 gpu): $(show $OUT/bench_rccl1_default.json)"
timeout -k 5 400 $B --no-cpu-baseline --gpus 1 --force-dist --workload c4_patch --gather full > $OUT/bench_rccl1_c4_patch_full.json 2> $OUT/bench_rccl1_c4_patch_full.err; echo "force-dist c4_patch full: $(show $OUT/bench_rccl1_c4_patch_full.json)"
cd /tmp
for WL in c5_track; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$WL -o trace --output-format csv -- python $REPO/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-verify > $OUT/trace_$WL.json 2> $OUT/trace_$WL.err
  python $REPO/scripts/kernel_stats.py $OUT/trace_$WL > $OUT/trace_${WL}_kernel_stats.txt 2>&1; head -40 $OUT/trace_${WL}_kernel_stats.txt | cut -c1-200
done
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_c2_dense_refresh -o trace --output-format csv -- python $REPO/bench.py --workload c2_dense --refresh-maps --steps 20 --warmup 3 --no-cpu-baseline --no-verify > $OUT/trace_c2_dense_refresh.json 2> $OUT/trace_c2_dense_refresh.err
python $REPO/scripts/kernel_stats.py $OUT/trace_c2_dense_refresh > $OUT/trace_c2_dense_refresh_kernel_stats.txt 2>&1; head -30 $OUT/trace_c2_dense_refresh_kernel_stats.txt | cut -c1-200
rm -rf $OUT/*/trace/*/*hip_api* 2>/dev/null
