"""Per-kernel statistics of a rocprofv3 --kernel-trace run: calls, average / median / min duration in microseconds.
usage: python scripts/kernel_stats.py <rocprof output dir> [substring filter ...]"""
import csv, glob, os, sys
from collections import defaultdict

root, filt = sys.argv[1], sys.argv[2:]
d = defaultdict(list)
for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-84s %6s %10s %10s %10s" % ("kernel", "calls", "avg us", "median us", "min us"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if filt and not any(s in k for s in filt):
        continue
    v2 = sorted(v)
    print("%-84s %6d %10.2f %10.2f %10.2f" % (k[:84], len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))
