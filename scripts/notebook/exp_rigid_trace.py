"""Kernel list of the tracking iteration (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from d3fields_amd import Fusion, rigid
dev = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rigid_tracking.npz"))
f = Fusion(num_cam=4, device="cuda:0")
f.curr_obs_torch = {k: torch.from_numpy(g[k]).to(dev) for k in ("depth", "K", "pose")}
f.curr_obs_torch["dino_feats"] = torch.from_numpy(g["in_dino_feats"]).to(dev)
f.H, f.W, f.mu = int(g["H"]), int(g["W"]), float(g["mu"])
src = torch.from_numpy(g["src_feats"]).to(dev)
last = torch.from_numpy(g["last_pts"]).to(dev)
rigid.track_rigid(f, src, last, use_graph=False, iters=20)
torch.cuda.synchronize()
