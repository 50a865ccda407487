"""GPU experiment: LDS texel windows (D3F_EXP_WINDOW) vs the cell-run gather on the patch-resolution workloads.
Bit-identity of every output against the default launch, and the fused kernel's time (d3f_profile_next_eval)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda:0")
workloads = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2_patch", "c3_patch"]
variants = [("default", {}),
            ("T64 U1 occ4", {"D3F_EXP_WINDOW": "64", "D3F_EXP_WINDOW_OCC": "4"}),
            ("T64 U1 occ3", {"D3F_EXP_WINDOW": "64", "D3F_EXP_WINDOW_OCC": "3"}),
            ("T128 U1 occ2", {"D3F_EXP_WINDOW": "128", "D3F_EXP_WINDOW_OCC": "2"}),
            ("T32 U1 occ4", {"D3F_EXP_WINDOW": "32", "D3F_EXP_WINDOW_OCC": "4"}),
            ("T64 U3 vc1", {"D3F_EXP_WINDOW": "64", "D3F_EXP_WINDOW_U": "3"}),
            ("T64 U3 vc2", {"D3F_EXP_WINDOW": "64", "D3F_EXP_WINDOW_U": "3", "D3F_EXP_WINDOW_VC": "2"}),
            ("T64 U3 vc1 again", {"D3F_EXP_WINDOW": "64", "D3F_EXP_WINDOW_U": "3"}),
            ("T64 U1 pool 16", {"D3F_EXP_WINDOW": "64", "D3F_EXP_WINDOW_POOL": "16"}),
            ("T64 U3 pool 8", {"D3F_EXP_WINDOW": "64", "D3F_EXP_WINDOW_U": "3", "D3F_EXP_WINDOW_POOL": "8"})]
KEYS = ("D3F_EXP_WINDOW", "D3F_EXP_WINDOW_OCC", "D3F_EXP_WINDOW_POOL", "D3F_EXP_WINDOW_U", "D3F_EXP_WINDOW_VC", "D3F_EXP_WINDOW_DEBUG")
for wl in workloads:
    for points in ("grid", "random"):
        f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1, points)
        f.cache_point_order = False
        ref = None
        for tag, env in variants:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            with torch.no_grad():
                fn = lambda: f.batch_eval(pts, return_names=names)
                try:
                    out = fn(); fn()
                    torch.cuda.synchronize()
                    t = bench.fused_kernel_time_ms(fn, 10, dev)
                except Exception as exc:
                    print("%s %s | %s: FAILED %r" % (wl, points, tag, exc), flush=True)
                    continue
            if ref is None:
                ref = {k: v.clone() for k, v in out.items()}
                same = "reference"
            else:
                same = "identical" if all(torch.equal(out[k], ref[k]) for k in ref) else \
                    "DIFFERENT " + ",".join(k for k in ref if not torch.equal(out[k], ref[k]))
            print("%s %s | %-10s kernel avg %.3f med %.3f min %.3f ms | %s" % (wl, points, tag, t[0], t[1], t[2], same), flush=True)
        del f, pts
        torch.cuda.empty_cache()
