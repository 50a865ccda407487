"""Condenses rocprofv3 csv output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print("== rocprofv3 --kernel-trace --stats ==")
for p in find("*kernel_stats.csv"):
    print("#", os.path.relpath(p, root))
    with open(p) as f:
        for i, row in enumerate(csv.reader(f)):
            if i < 12:
                print(",".join(c[:110] for c in row))

print("\n== per-kernel durations from kernel_trace (ns) ==")
for p in find("trace*kernel_trace.csv"):
    d = defaultdict(list)
    with open(p) as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
        v2 = sorted(v)
        print("%-90s n=%4d avg=%10.0f med=%10.0f min=%10.0f" % (k[:90], len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))

print("\n== PMC passes (per dispatch averages of the dominant kernel) ==")
for p in find("pmc*counter_collection.csv"):
    agg = defaultdict(lambda: defaultdict(list))
    with open(p) as f:
        for r in csv.DictReader(f):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "fused_eval" not in k and "pairwise" not in k:
            continue
        for c, v in cs.items():
            print("%-56s %-20s n=%4d avg=%16.1f" % (k[:56], c, len(v), sum(v) / len(v)))
