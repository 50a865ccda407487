#!/bin/bash
# round 3, session 50: stability of the in-kernel waits of d3f_track_run -- the equality tests 15 times over, and 300 frames of the tracker
set -u
for i in $(seq 1 15); do timeout -k 5 120 python -m pytest tests/test_gpu_callers.py -m gpu -q -x -k "track_run" 2>&1 | tail -1; done | sort | uniq -c
timeout -k 5 200 python - <<'PY'
import numpy as np, torch, time, os, sys
sys.path.insert(0, os.getcwd())
from d3fields_amd import Fusion, rigid
dev = torch.device("cuda:0")
g = np.load("tests/golden/rigid_tracking.npz")
ft = Fusion(num_cam=4, device="cuda:0")
ft.curr_obs_torch = {k: torch.from_numpy(g[k]).to(dev) for k in ("depth", "K", "pose")}
ft.curr_obs_torch["dino_feats"] = torch.from_numpy(g["in_dino_feats"]).to(dev)
ft.H, ft.W, ft.mu = int(g["H"]), int(g["W"]), float(g["mu"])
n = int(g["n"])
src = torch.from_numpy(g["src_feats"]).to(dev)
last = torch.from_numpy(np.stack([p for p in g["last_pts"]])).to(dev)
tr = rigid.RigidTracker(ft, 2, n)
assert tr.loop
ref = None; bad = 0
torch.cuda.synchronize(); t0 = time.perf_counter()
for frame in range(300):
    cur, loss = tr.run(ft, src, last)
    if ref is None: ref = cur.clone()
    elif not torch.equal(cur, ref) or not torch.isfinite(loss): bad += 1
torch.cuda.synchronize()
print("300 frames: %.2f ms per frame, %d frames differing from the first, max |got - golden| = %.2e" % ((time.perf_counter() - t0) * 1e3 / 300, bad, np.abs(ref.view(2, n, 3).cpu().numpy() - g["match_pts"]).max()))
PY
