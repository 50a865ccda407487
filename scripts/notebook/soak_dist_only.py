"""Soak of the distance-only kernel (fuse_direct.hip: fused_eval_dist_kernel) against the CPU oracle: random view counts (1-8), map sizes that are
no multiples of the 4 x 8 depth tiles, smooth / noise depth maps, clouds at three scales with non-finite points, below and above the 2^22-point
threshold of the tiled lookups, both query modes.  Prints one line per case; exits non-zero on the first mismatch.
    python scripts/notebook/soak_dist_only.py [cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d3fields_amd import Fusion, synth      # noqa: E402
from oracle import c_oracle as O            # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
r = np.random.default_rng(2026)
for c in range(cases):
    V = int(r.integers(1, 9))
    H, W = int(r.integers(17, 300)), int(r.integers(19, 400))
    kind = "smooth" if r.integers(0, 3) else "stress"
    big = c % 2 == 1
    n = int(r.integers(1 << 22, (1 << 22) + 300000)) if big else int(r.integers(1, 200000))
    sc = synth.make_scene(V, H, W, kind, seed=c)
    pts_c = synth.random_cloud(n, seed=100 + c) * float(r.choice([0.3, 1.0, 3.0]))
    for j in r.integers(0, n, 6):
        pts_c[j, int(r.integers(0, 3))] = float(r.choice([np.inf, -np.inf, np.nan, 0.0, 1e30, 1e-40]))
    f = Fusion(num_cam=V, device=str(dev))
    f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
    f.H, f.W = H, W
    f.record_plans = True
    pts = pts_c.to(dev)
    with torch.no_grad():
        a = f.batch_eval(pts, return_names=[])
        plan = f.last_plan()["kernel"]
        d = f.eval_dist(pts)
    ref = O.eval_field(sc["depth"], sc["K"], sc["pose"], pts_c, [])
    ref_d = O.eval_field(sc["depth"], sc["K"], sc["pose"], pts_c, [], mode="eval_dist")
    ok = (np.array_equal(a["valid_mask"].cpu().numpy(), ref["valid_mask"].astype(bool)) and np.array_equal(a["dist"].cpu().numpy(), ref["dist"], equal_nan=True) and
          np.array_equal(d["valid_mask"].cpu().numpy(), ref_d["valid_mask"].astype(bool)) and np.array_equal(d["dist"].cpu().numpy(), ref_d["dist"], equal_nan=True))
    print("case %2d V=%d %dx%d %-6s n=%8d %-48s %s" % (c, V, H, W, kind, n, plan, "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        sys.exit(1)
print("all %d cases bit-identical to the oracle" % cases)
