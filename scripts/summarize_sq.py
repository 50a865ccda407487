"""Condenses the SQ / LDS / TCP counter passes of scripts/r6_counters.sh into one table per kernel.

Every busy fraction is taken against the cycles the kernel really ran: cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over
the eight XCDs) of the SAME pass, clock = cycles / the dispatch's duration in that pass -- never an assumed 2.4 GHz.
Units (MI355X_MICROARCH.md, cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over the
waves; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS-array cycles summed over the CUs; SQ_INSTS_* count wave instructions.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
NCU, NSIMD, NXCD = 256, 1024, 8
KEEP = ("fused_eval", "pairwise", "gate_probe")

per_kernel = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values (one per dispatch)
dur = defaultdict(list)                                  # kernel -> dispatch durations (ns) of the counter passes
for p in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(p) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if not any(s in k for s in KEEP):
                continue
            per_kernel[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and "End_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))


def mean(v):
    return sum(v) / len(v) if v else float("nan")


for k, cs in sorted(per_kernel.items(), key=lambda kv: -mean(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    c = {n: mean(v) for n, v in cs.items()}
    n_disp = len(cs.get("SQ_INSTS_VALU", cs.get("GRBM_GUI_ACTIVE", [])))
    print("== %s" % k[:150])
    for n in sorted(c):
        print("   %-34s %16.0f  (n=%d)" % (n, c[n], len(cs[n])))
    cyc = c.get("GRBM_GUI_ACTIVE", float("nan")) / NXCD
    d = mean(dur[k])
    print("   -- derived (per dispatch; cycles = GRBM_GUI_ACTIVE / 8 = %.0f; duration under the profiler %.1f us -> clock %.2f GHz)" %
          (cyc, d / 1e3, cyc / d if d == d and d > 0 else float("nan")))

    def frac(name, num, den):
        if num == num and den == den and den > 0:
            print("   %-58s %6.3f" % (name, num / den))

    frac("LDS array busy    = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles)", c.get("SQ_LDS_IDX_ACTIVE", float("nan")), NCU * cyc)
    frac("LDS bank conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE", c.get("SQ_LDS_BANK_CONFLICT", float("nan")), c.get("SQ_LDS_IDX_ACTIVE", float("nan")))
    frac("VALU busy         = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles)", 4 * c.get("SQ_ACTIVE_INST_VALU", float("nan")), NSIMD * cyc)
    frac("MFMA pipe busy    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)", c.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")), NSIMD * cyc)
    frac("LDS issue busy    = 4 x SQ_ACTIVE_INST_LDS / (1024 x cycles)", 4 * c.get("SQ_ACTIVE_INST_LDS", float("nan")), NSIMD * cyc)
    frac("VMEM issue busy   = 4 x SQ_ACTIVE_INST_VMEM / (1024 x cycles)", 4 * c.get("SQ_ACTIVE_INST_VMEM", float("nan")), NSIMD * cyc)
    frac("scalar busy       = 4 x SQ_ACTIVE_INST_SCA / (1024 x cycles)", 4 * c.get("SQ_ACTIVE_INST_SCA", float("nan")), NSIMD * cyc)
    frac("waves parked      = SQ_WAIT_ANY / SQ_WAVE_CYCLES", c.get("SQ_WAIT_ANY", float("nan")), c.get("SQ_WAVE_CYCLES", float("nan")))
    frac("waves issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES", c.get("SQ_WAIT_INST_ANY", float("nan")), c.get("SQ_WAVE_CYCLES", float("nan")))
    frac("waves issuing     = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES", c.get("SQ_ACTIVE_INST_ANY", float("nan")), c.get("SQ_WAVE_CYCLES", float("nan")))
    frac("LDS issue stalls  = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES", c.get("SQ_WAIT_INST_LDS", float("nan")), c.get("SQ_WAVE_CYCLES", float("nan")))
    frac("waves per SIMD (resident) = 4 x SQ_WAVE_CYCLES / (1024 x cycles)", 4 * c.get("SQ_WAVE_CYCLES", float("nan")), NSIMD * cyc)
    frac("CUs busy          = 4 x SQ_BUSY_CU_CYCLES / (256 x cycles)", 4 * c.get("SQ_BUSY_CU_CYCLES", float("nan")), NCU * cyc)
    frac("VALU cycles per instruction = 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU", 4 * c.get("SQ_ACTIVE_INST_VALU", float("nan")), c.get("SQ_INSTS_VALU", float("nan")))
    frac("vector-L1 clocked (TCP_GATE_EN1_sum / (256 x cycles); a clock enable, ~1 for every kernel: not a utilisation)", c.get("TCP_GATE_EN1_sum", float("nan")), NCU * cyc)
    frac("TA busy           = TA_TA_BUSY_sum / (256 x cycles)", c.get("TA_TA_BUSY_sum", float("nan")), NCU * cyc)
    frac("TD busy           = TD_TD_BUSY_sum / (256 x cycles)", c.get("TD_TD_BUSY_sum", float("nan")), NCU * cyc)
    frac("L1 pending stall  = TCP_PENDING_STALL_CYCLES_sum / (256 x cycles)", c.get("TCP_PENDING_STALL_CYCLES_sum", float("nan")), NCU * cyc)
    frac("L1->L2 read requests per L1 access = TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum", c.get("TCP_TCC_READ_REQ_sum", float("nan")), c.get("TCP_TOTAL_CACHE_ACCESSES_sum", float("nan")))
    frac("L1 accesses per cycle per CU = TCP_TOTAL_CACHE_ACCESSES_sum / (256 x cycles)", c.get("TCP_TOTAL_CACHE_ACCESSES_sum", float("nan")), NCU * cyc)
    frac("L2 read latency (cycles) = TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum", c.get("TCP_TCC_READ_REQ_LATENCY_sum", float("nan")), c.get("TCP_TCC_READ_REQ_sum", float("nan")))
    print()
