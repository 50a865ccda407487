"""How long does the HOST need per Fusion.batch_eval call (Python + ctypes + launches), vs the device time of the step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda:0")
for wl in ("c2_patch", "c2_dense"):
    f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1)
    for mode in ("cache_off_async", "cache_off_sync", "cache_on"):
        f.cache_point_order = mode == "cache_on"
        f.async_probes = mode != "cache_off_sync"
        with torch.no_grad():
            for _ in range(5):
                f.batch_eval(pts, return_names=names)
            torch.cuda.synchronize()
            # host time per call when the GPU is NOT the limiter: tiny query of the same code path
            small = pts[:70000].contiguous()
            for _ in range(3):
                f.batch_eval(small, return_names=names)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                f.batch_eval(small, return_names=names)
            t_issue = (time.perf_counter() - t0) / 200
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                f.batch_eval(pts, return_names=names)
            torch.cuda.synchronize()
            t_step = (time.perf_counter() - t0) / 50
        print("%-9s %-16s host issue %.1f us per call (70 k-point query), full-size step %.1f us" % (wl, mode, 1e6 * t_issue, 1e6 * t_step))
