import os, sys, ctypes
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import torch, bench
from d3fields_amd import _lib
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "c2_dense"
f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1, "grid")
lib = ctypes.CDLL(_lib.library_path())
buf = (ctypes.c_ulonglong * 8)()
with torch.no_grad():
    for _ in range(3):
        f.batch_eval(pts, return_names=names)
    lib.d3f_debug_phase_ticks(buf, 1)
    R = 10
    for _ in range(R):
        f.batch_eval(pts, return_names=names)
    lib.d3f_debug_phase_ticks(buf, 1)
n = buf[7]
names_ = ["krt+tile decode", "phase A (pts, depth, records)", "per-point sums", "gather point 1 (2 rounds + store)", "gather point 2", "", ""]
print("%s: %d workgroups over %d launches" % (wl, n, R))
tot = 0
for k in range(5):
    us = buf[k] * 0.01 / max(n, 1)
    tot += us
    print("  %-36s %.2f us per workgroup" % (names_[k], us))
print("  total %.2f us" % tot)
