"""debug / timing of d3f_track_run (all optimiser steps of a frame in one launch) against d3f_track_step per iteration:
    python scripts/exp_track_run.py
prints, for a few frames each, the final parameters and the distance to the golden keypoints."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from d3fields_amd import Fusion, rigid, _lib
dev = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rigid_tracking.npz"))
ft = Fusion(num_cam=4, device="cuda:0")
ft.curr_obs_torch = {k: torch.from_numpy(g[k]).to(dev) for k in ("depth", "K", "pose")}
ft.curr_obs_torch["dino_feats"] = torch.from_numpy(g["in_dino_feats"]).to(dev)
ft.H, ft.W, ft.mu = int(g["H"]), int(g["W"]), float(g["mu"])
n = int(g["n"])
src = torch.from_numpy(g["src_feats"]).to(dev)
last = torch.from_numpy(np.stack([p for p in g["last_pts"]])).to(dev)
for mode in ("step", "run-eager", "run-graph"):
    tr = rigid.RigidTracker(ft, 2, n, loop_launch=(mode != "step"))
    for frame in range(3):
        if mode == "run-eager":
            with torch.no_grad():
                for k, t in tr.shadow.curr_obs_torch.items():
                    t.copy_(ft.curr_obs_torch[k])
                tr.last.copy_(last); tr.src.copy_(src)
            tr._rewind()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            cur, loss = tr._step(tr.iters)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        else:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            cur, loss = tr.run(ft, src, last)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        got = cur.view(2, n, 3).cpu().numpy()
        print("%-10s frame %d: %.2f ms, max |got - golden| = %.2e m, t = %s, w = %s, adam step = %s, loss = %.6f, scratch tail = %s" % (
            mode, frame, dt * 1e3, np.abs(got - g["match_pts"]).max(), tr.t_params.flatten().tolist()[:3], tr.log_r.flatten().tolist()[:3],
            tr.state[2 * 12:2 * 13].tolist(), float(loss), tr.scratch[-4:].view(torch.int32).tolist()), flush=True)
