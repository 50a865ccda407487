import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3fields_amd import Fusion, synth
dev = torch.device("cuda:0")
V, H, W = 4, 480, 640
sc = synth.make_scene(V, H, W, "smooth")
f = Fusion(num_cam=V, device="cuda:0")
f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
f.curr_obs_torch["dino_feats"] = synth.random_map(V, 48, 64, 384, seed=1, device=dev)
f.H, f.W = H, W
src = synth.random_cloud(500, seed=9).to(dev)
with torch.no_grad():
    for _ in range(20):
        f.eval(src, return_names=["dino_feats"])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        f.eval(src, return_names=["dino_feats"])
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
