#!/bin/bash
# round 4, session 22 (EXPERIMENTS builds with -DD3F_WIN_ABLATE=bits), split launch on: the gather kernel's skeleton (25 = no later
# copies, no corner reads, no arithmetic; 29 = also no HBM writes), non-temporal row stores (32), and the box's store-only floor
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4v; mkdir -p $OUT
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1 D3F_EXP_WINDOW_SPLIT=1
python $REPO/scripts/fill_floor.py 2>&1 | grep fill
cd /tmp
for AB in 0 25 29 32 0; do
  cp $REPO/build_ab/ablate_$AB.so $REPO/d3fields_amd/libd3fields_hip.so
  for WL in c2_patch c4_patch; do
    rm -rf $OUT/prof_${AB}_$WL
    timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/prof_${AB}_$WL -o p -- python $REPO/bench.py --no-cpu-baseline --no-verify --steps 20 --workload $WL > $OUT/prof_${AB}_$WL.log 2>&1
    python - $OUT/prof_${AB}_$WL $AB $WL <<'PY'
import sqlite3,glob,sys
db=glob.glob(sys.argv[1]+'/*.db')[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%window%' group by s.kernel_name order by 3 desc"
print("ablate", sys.argv[2], sys.argv[3], " | ".join("%s n=%d avg %.1f min %.1f us" % (("setup" if "setup" in r[0] else "gather"), r[1], r[2]/1e3, r[3]/1e3) for r in c.execute(q)))
PY
  done
done
