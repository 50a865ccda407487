"""What the thin maps riding along cost: ref_patch / c3_patch with and without them (HIP events around batch_eval, median of 30)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda:0")
for wl in ("ref_patch", "c3_patch", "c3_dense"):
    f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1)
    for sel in (names, names[:1], names[1:]):
        with torch.no_grad():
            for _ in range(5): f.batch_eval(pts, return_names=sel)
            torch.cuda.synchronize()
            ts = []
            for _ in range(30):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f.batch_eval(pts, return_names=sel); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
        ts.sort()
        print("%-10s %-40s %.3f ms" % (wl, ",".join(sel), ts[len(ts) // 2]), flush=True)
    del f, pts
    torch.cuda.empty_cache()
