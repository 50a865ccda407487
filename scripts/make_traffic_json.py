"""Rebuilds profiles/traffic.json from the rocprofv3 summaries of scripts/r6_profile_all.sh.
usage: python scripts/make_traffic_json.py profiles/r2_v3 [more dirs ...]   (later dirs override earlier ones)
traffic = FETCH_SIZE[KB]*1024*2 + WRITE_SIZE[KB]*1024 per launch of the dominant fused_eval kernel (see _comment)."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402
out_path = os.path.join(ROOT, "profiles", "traffic.json")
try:
    data = json.load(open(out_path))
except (OSError, ValueError):
    data = {}
data["_comment"] = ("HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (separate --pmc runs, kernel-trace only; "
                    "scripts/r6_profile_all.sh, rebuilt by scripts/make_traffic_json.py). traffic = FETCH_SIZE[KB]*1024*2 (gfx950 tallies "
                    "16-B/lane coalesced reads at half size, MI355X_MICROARCH.md HBM section; confirmed here: FETCH_SIZE*1024 == "
                    "TCC_EA0_RDREQ_sum*64 with TCC_EA0_RDREQ_32B_sum == 0) + WRITE_SIZE[KB]*1024. FETCH_SIZE counts Infinity-Cache hits "
                    "(fabric traffic, not DRAM traffic). bench.py copies the entry of its workload into roofline.traffic "
                    "(traffic_measured_in_run: false) -- only when the run has the same point set (`<workload>`: its grid, "
                    "`<workload>_random`: the cloud) and the same number of points as the profiled launch (`points`).")
for d in sys.argv[1:]:
    for path in sorted(glob.glob(os.path.join(d, "*_summary.txt"))):
        wl = os.path.basename(path)[:-len("_summary.txt")]
        text = open(path).read()
        dur = re.search(r"^(?:void )?d3f::(fused_eval\w*kernel(?:<[^>]*>)?)\(d3f::EvalParams\)\s+n=\s*(\d+) avg=\s*(\d+) med=\s*(\d+)", text, re.M)
        if not dur:
            continue
        c = {}
        dominant = re.match(r"fused_eval\w*kernel", dur.group(1)).group(0)      # a gated cloud launches two fused kernels: the winner's counters
        for m in re.finditer(r"(fused_eval\w*kernel)\S*(?: \S+)*?\s+([A-Z][A-Z0-9_a-z]+)\s+n=\s*\d+ avg=\s*([0-9.]+)\s*$", text, re.M):
            if m.group(1) == dominant:
                c[m.group(2)] = float(m.group(3))
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        e = {"kernel": dur.group(1), "kernel_avg_ns": int(dur.group(3)), "kernel_median_ns": int(dur.group(4)),
             "fetch_size_kb": round(c["FETCH_SIZE"], 1), "write_size_kb": round(c["WRITE_SIZE"], 1),
             "traffic_bytes": int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)}
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if "TCC_EA0_RDREQ_sum" in c and "TCC_EA0_RDREQ_LEVEL_sum" in c and c["TCC_EA0_RDREQ_sum"]:
            e["ea_read_latency_cycles"] = round(c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"], 1)
        if "SQ_INSTS_VMEM_RD" in c:
            e["wave_loads"] = int(c["SQ_INSTS_VMEM_RD"])
        if "SQ_INSTS_VALU" in c:
            e["valu_insts"] = int(c["SQ_INSTS_VALU"])
        e["source"] = os.path.relpath(path, ROOT)
        fp = re.search(r"^source_fingerprint: ([0-9a-f]{64})", text, re.M)
        if fp:          # the sources the profiled library was built from: bench.py prints the entry only for the same library
            e["source_fingerprint"] = fp.group(1)
        # the launch the counters belong to: bench.py prints the entry only for a run with the same point set and count
        # (`<workload>` = its grid, `<workload>_random` = the cloud of --points random)
        base = wl
        for suffix in ("_random", "_surface"):
            if wl.endswith(suffix):
                base = wl[:-len(suffix)]
        if wl.endswith("_surface"):          # the surface cloud's size depends on the scene: taken from the bench line of the traced run
            m = re.search(r'"points_per_gpu": (\d+)', open(path[:-len("_summary.txt")] + ".bench_trace.json").read()) if os.path.exists(path[:-len("_summary.txt")] + ".bench_trace.json") else None
            if m:
                e["points"] = int(m.group(1))
        elif base in WORKLOADS:
            e["points"] = int(WORKLOADS[base].get("N_cloud", WORKLOADS[base]["N"]) if wl.endswith("_random") else WORKLOADS[base]["N"])
        data[wl] = e
json.dump(data, open(out_path, "w"), indent=1)
print("wrote", out_path, sorted(k for k in data if not k.startswith("_")))
