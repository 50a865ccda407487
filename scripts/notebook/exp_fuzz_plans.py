import sys, collections
import os; R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import test_gpu_fuzz as T
from d3fields_amd import Fusion, create_init_grid, synth
dev = torch.device("cuda:0")
cnt = collections.Counter()
for seed in range(40):
    c = T._case(seed)
    V, H, W = c["V"], c["H"], c["W"]
    sc = synth.make_scene(V, H, W, c["kind"])
    maps, names = {}, []
    if c["C"]:
        maps["dino_feats"] = synth.random_map(V, c["fhw"][0], c["fhw"][1], c["C"], seed=seed + 1, device=dev); names.append("dino_feats")
    if c["NI"]:
        maps["mask"] = synth.random_onehot_mask(V, H, W, c["NI"], seed=seed + 2, device=dev); names.append("mask")
    if c["color"]:
        maps["color_tensor"] = torch.rand(V, H, W, 3, device=dev); names.append("color_tensor")
    f = Fusion(num_cam=V, device=str(dev)); f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}; f.curr_obs_torch.update(maps); f.H, f.W = H, W; f.record_plans = True
    r = np.random.default_rng(2000 + seed)
    if c["lattice"]:
        dims = [(48, 44, 32), (130, 9, 60), (64, 64, 17), (20, 120, 28)][int(r.integers(0, 4))]; step = float(r.choice([0.004, 0.0107, 0.02]))
        box = dict(x_lower=-dims[0] * step / 2, x_upper=dims[0] * step / 2 - step / 4, y_lower=-dims[1] * step / 2, y_upper=dims[1] * step / 2 - step / 4, z_lower=-0.2, z_upper=-0.2 + dims[2] * step - step / 4)
        pts = create_init_grid(box, step)[0].to(dev)
    else:
        pts = synth.random_cloud(70001, seed=seed).to(dev)
    with torch.no_grad(): f.batch_eval(pts, return_names=names)
    p = f.last_plan(); cnt[p.get("kernel")] += 1
    print(seed, c["V"], c["C"], c["fhw"], c["NI"], c["color"], "lattice" if c["lattice"] else "cloud", p.get("kernel"), p.get("point_order"))
print(cnt)
