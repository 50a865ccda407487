"""GPU experiment: tile size x XCD remap x occupancy throttle, per workload (one process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda:0")
for wl in sys.argv[1:]:
    f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1)

    def t_ms(flags, reps=6):
        f.tuning_flags = flags
        with torch.no_grad():
            for _ in range(2):
                f.batch_eval(pts, return_names=names)
            ev = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f.batch_eval(pts, return_names=names); b.record(); ev.append((a, b))
            torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        return ts[len(ts) // 2]

    print(wl, "auto: %.3f" % t_ms(0), flush=True)
    for tl in (2, 3, 4):
        print("  tile %d | " % (1 << tl) + " | ".join("chunk %d: %.3f" % (1024 << (k - 1), t_ms((tl << 8) | (k << 29))) for k in (2, 3, 4, 5)), flush=True)
    del f, pts
    torch.cuda.empty_cache()
