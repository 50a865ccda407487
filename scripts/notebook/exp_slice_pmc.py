"""Run a few evals of a [..., :cs] channel slice of the dense map with given tuning flags (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3fields_amd import Fusion, create_init_grid, synth

cs, flags = int(sys.argv[1]), int(sys.argv[2], 0)
dev = torch.device("cuda:0")
V, H, W, C = 4, 480, 640, 384
sc = synth.make_scene(V, H, W, "smooth")
f = Fusion(num_cam=V, device="cuda:0")
f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
feats = synth.random_map(V, H, W, C, seed=1, device=dev)
f.curr_obs_torch["slice"] = feats[..., :cs]
f.H, f.W = H, W
pts, _ = create_init_grid(synth.WORK_BOX, 0.005)
pts = pts.to(dev)
f.tuning_flags = flags
with torch.no_grad():
    for _ in range(4):
        f.batch_eval(pts, return_names=["slice"])
torch.cuda.synchronize()
