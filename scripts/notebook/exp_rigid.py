"""GPU measurement: Fusion.rigid_tracking, eager launches vs one HIP graph replayed; difference to the reference's result."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from d3fields_amd import Fusion

dev = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rigid_tracking.npz"))
f = Fusion(num_cam=4, device="cuda:0")
f.curr_obs_torch = {k: torch.from_numpy(g[k]).to(dev) for k in ("depth", "K", "pose")}
f.curr_obs_torch["dino_feats"] = torch.from_numpy(g["in_dino_feats"]).to(dev)
f.H, f.W, f.mu = int(g["H"]), int(g["W"]), float(g["mu"])
n = int(g["n"])
info = {"a": {"src_feats": torch.from_numpy(g["src_feats"][:n])}, "b": {"src_feats": torch.from_numpy(g["src_feats"][n:])}}
last = [p for p in g["last_pts"]]
for use_graph, fused, whole in ((False, False, False), (True, False, False), (True, False, False), (True, True, False), (True, True, False), (True, True, False),
                               (True, True, True), (True, True, True), (True, True, True), (True, True, True)):
    f.use_hip_graph, f.fused_tracking, f.graph_whole_tracking_loop = use_graph, fused, whole
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = f.rigid_tracking(info, last, None, n)
    dt = time.perf_counter() - t0
    got = np.stack(res["match_pts_list"])
    print("graph=%s fused=%s whole-loop=%s: %.1f ms per call (100 iterations), max |got - reference| = %.3e m, max |got - true| = %.4f m"
          % (use_graph, fused, whole, dt * 1e3, np.abs(got - g["match_pts"]).max(), np.abs(got - g["true_pts"]).max()), flush=True)
