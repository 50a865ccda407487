#!/bin/bash
# round 3, session 36: effective L2 capacity of one XCD for strided items (scripts/microbench/l2_probe.hip), timing + TCC hit / miss counters
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3_l2probe; mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/microbench/l2_probe.hip -o /tmp/l2_probe 2> /dev/null
timeout -k 5 300 /tmp/l2_probe > $OUT/l2_probe_timing.txt 2>&1; cat $OUT/l2_probe_timing.txt | head -90
cd /tmp; timeout -k 5 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $OUT/pmc -o pmc --output-format csv -- /tmp/l2_probe > $OUT/l2_probe_under_pmc.txt 2> $OUT/pmc.err; cd $REPO
python - <<'PY' > $OUT/l2_probe_hit_rates.txt
import csv, glob
rows = []
for p in glob.glob("gpurun_out/r3_l2probe/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "probe" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
by = {}
for d, c, v in rows:
    by.setdefault(d, {})[c] = v
cfgs = [(256, 256), (256, 1536), (512, 1536), (128, 1536), (512, 4096), (256, 4096), (512, 512), (1536, 1536)]
ns = [1024, 2048, 4096, 6144, 8192, 10240, 12288, 16384, 24576, 32768]
disp = sorted(by)
k = 0
print("%-22s %8s %10s %12s %12s %10s" % ("item B @ stride B", "items", "set KiB", "hits", "misses", "hit rate"))
for item, stride in cfgs:
    for n in ns:
        if n * stride > (1 << 30):
            continue
        if 2 * k + 1 >= len(disp):
            break
        c = by[disp[2 * k + 1]]            # the timed launch (40 passes); the one before it warmed the L2s
        h, m = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        print("%6d @ %-13d %8d %10.0f %12.0f %12.0f %9.1f%%" % (item, stride, n, n * item / 1024, h, m, 100 * h / max(h + m, 1)))
        k += 1
PY
cat $OUT/l2_probe_hit_rates.txt | head -90
