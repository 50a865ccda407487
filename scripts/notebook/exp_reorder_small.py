"""GPU experiment: launch geometry for the Morton walk when the maps are SMALL (patch-res features, mask):
random clouds need the walk for L1/L2 locality, but the 8-point / XCD-eighth geometry was tuned on dense maps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda:0")
FOR, XCD, NOR = 1 << 14, 1 << 12, 1 << 13
for wl in sys.argv[1:]:
    for points in ("random", "grid"):
        f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1, points)

        def k_ms(flags):
            f.tuning_flags = flags
            with torch.no_grad():
                fn = lambda: f.batch_eval(pts, return_names=names)
                fn(); fn()
                return bench.fused_kernel_time_ms(fn, 6, dev)[1]

        print(wl, points, "auto: %.3f  noreorder: %.3f" % (k_ms(0), k_ms(NOR)), flush=True)
        for tl in (3, 4, 5, 6, 7):
            print("  walk tile %3d | xcd-eighths: %.3f | round-robin: %.3f | chunk 2k tiles: %.3f | chunk 8k tiles: %.3f"
                  % (1 << tl, k_ms(FOR | (tl << 8)), k_ms(FOR | XCD | (tl << 8)), k_ms(FOR | (tl << 8) | (2 << 29)),
                     k_ms(FOR | (tl << 8) | (4 << 29))), flush=True)
        del f, pts
        torch.cuda.empty_cache()
