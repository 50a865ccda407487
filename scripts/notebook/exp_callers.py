"""GPU measurement of the caller-side kernels: farthest point sampling (3-D and pixel variants), the grid-shell
pre-filter, the distance-only query of the 1-mm grid, the backward of eval, one rigid-tracking frame.  Device time from HIP
events around the call (all of it is asynchronous on the current stream); run under rocprofv3 for the per-kernel breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from d3fields_amd import Fusion, fps, synth, pcd_utils, create_init_grid

dev = torch.device("cuda:0")


def dev_ms(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


which = set(sys.argv[1:]) or {"fps", "shell", "dist", "backward", "rigid"}
if "fps" in which:
    cloud = torch.from_numpy(np.random.default_rng(5).normal(size=(200000, 3)).astype(np.float32)).to(dev)
    print("fps 100 of 200000 points: %.3f ms" % dev_ms(lambda: fps(cloud, 100, init_idx=0)), flush=True)
    print("fps 1000 of 200000 points: %.3f ms" % dev_ms(lambda: fps(cloud, 1000, init_idx=0)), flush=True)
    small = cloud[:3000].contiguous()
    print("fps 100 of 3000 points: %.3f ms" % dev_ms(lambda: fps(small, 100, init_idx=0)), flush=True)
    mask = np.random.default_rng(9).random((480, 640)) < 0.3
    pix = np.array(mask.nonzero()).T
    t0 = time.perf_counter(); pcd_utils.fps_pixels(pix, 100, init_idx=3); torch.cuda.synchronize()
    print("fps_pixels 100 of %d pixels (incl. host transfers): %.3f ms wall" % (pix.shape[0], (time.perf_counter() - t0) * 1e3), flush=True)
f, pts, names, w, sc = bench.build_workload("c2_patch", dev, 0, 1)
if "shell" in which or "dist" in which:
    box, step = synth.WORK_BOX, 0.001
    with torch.no_grad():
        if "dist" in which:
            print("eval_grid dist-only, 1-mm grid (123.2 M points): %.3f ms" % dev_ms(lambda: f.eval_grid(box, step, return_names=[]), reps=3, warm=1), flush=True)
        if "shell" in which:
            print("grid_shell, 1-mm grid: %.3f ms" % dev_ms(lambda: f.grid_shell(box, step), reps=3, warm=1), flush=True)
if "backward" in which:
    q = pts[:100000].clone().requires_grad_(True)
    def fb():
        out = f.eval(q, return_names=["dino_feats"])
        (out["dino_feats"].sum() + out["dist"].sum()).backward()
    print("eval + backward, 100 k points x 384 channels (patch maps): %.3f ms" % dev_ms(fb), flush=True)
if "rigid" in which:
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rigid_tracking.npz"))
    ft = Fusion(num_cam=4, device="cuda:0")
    ft.curr_obs_torch = {k: torch.from_numpy(g[k]).to(dev) for k in ("depth", "K", "pose")}
    ft.curr_obs_torch["dino_feats"] = torch.from_numpy(g["in_dino_feats"]).to(dev)
    ft.H, ft.W, ft.mu = int(g["H"]), int(g["W"]), float(g["mu"])
    n = int(g["n"])
    info = {"a": {"src_feats": torch.from_numpy(g["src_feats"][:n])}, "b": {"src_feats": torch.from_numpy(g["src_feats"][n:])}}
    last = [p for p in g["last_pts"]]
    for rep in range(12):
        ft.single_launch_tracking = rep >= 4
        ft.loop_launch_tracking = rep >= 8
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = ft.rigid_tracking(info, last, None, n)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        got = np.stack(res["match_pts_list"])
        print("rigid_tracking frame (100 iterations, %d keypoints x 2 instances, %s): %.2f ms wall, max |got - reference| = %.2e m"
              % (n, "ONE launch per frame" if ft._tracker.loop else ("1 launch per step" if ft.single_launch_tracking else "5 launches per step"),
                 dt * 1e3, np.abs(got - g["match_pts"]).max()), flush=True)
