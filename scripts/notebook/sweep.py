"""GPU sweep: kernel time of the field query across workloads / return_names (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from d3fields_amd import synth

dev = torch.device("cuda:0")


def t_ms(fn, reps=10):
    with torch.no_grad():
        for _ in range(3):
            fn()
        ev = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); ev.append((a, b))
        torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for wl in sys.argv[1:] or ["c2_dense", "c2_patch", "c3_dense", "c3_patch", "c4_patch"]:
    f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1)
    n = pts.shape[0]
    f.curr_obs_torch["color_tensor"] = torch.rand(w["V"], w["H"], w["W"], 3, device=dev)
    if "mask" not in f.curr_obs_torch:
        f.curr_obs_torch["mask"] = synth.random_onehot_mask(w["V"], w["H"], w["W"], 8, seed=2, device=dev)
    rows = []
    for rn in (names, [], ["mask"], ["color_tensor"], ["dino_feats", "mask", "color_tensor"]):
        ms = t_ms(lambda: f.batch_eval(pts, return_names=rn))
        rows.append("%s: %.3f ms (%.3g pts/s)" % ("+".join(rn) or "dist-only", ms, n / ms * 1e3))
    ms = t_ms(lambda: f.eval_dist(pts))
    rows.append("eval_dist: %.3f ms" % ms)
    ms = t_ms(lambda: f.eval(pts[:60000], return_names=names, return_inter=True))
    rows.append("eval60k+inter: %.3f ms" % ms)
    print(wl, "N=%d" % n, " | ".join(rows), flush=True)
    del f, pts
    torch.cuda.empty_cache()
