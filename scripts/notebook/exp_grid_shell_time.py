"""Times Fusion.grid_shell (the keypoint pre-filter of select_features_rand, fusion.py:1418-1444) on bench.py's 1-mm work box: the pass over
123.2 M grid points + the compaction.  HIP events around the whole call (three launches + a scan), best of 5."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d3fields_amd import Fusion, synth      # noqa: E402

dev = torch.device("cuda:0")
for V, H, W in ((4, 480, 640), (8, 720, 1280)):
    sc = synth.make_scene(V, H, W, "smooth")
    f = Fusion(num_cam=V, device=str(dev))
    f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
    f.H, f.W = H, W
    for step in (0.001, 0.002):
        best, cnt = 1e9, 0
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            idx, pts = f.grid_shell(dict(synth.WORK_BOX), step, 0.005)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                best = min(best, e0.elapsed_time(e1))
            cnt = idx.numel()
        print("V=%d %dx%d step %.0f mm: grid_shell %.3f ms (host + device, best of 5), %d survivors" % (V, H, W, step * 1e3, best, cnt))
