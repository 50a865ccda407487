"""GPU measurement: rigid-tracking-style iteration (eval with grad + backward) at small N."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3fields_amd import Fusion, synth

dev = torch.device("cuda:0")
V, H, W = 4, 480, 640
sc = synth.make_scene(V, H, W, "smooth")
f = Fusion(num_cam=V, device="cuda:0")
f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
f.curr_obs_torch["dino_feats"] = synth.random_map(V, 48, 64, 1024, seed=1, device=dev)
f.H, f.W = H, W
for n in (300, 800, 5000):
    src = synth.random_cloud(n, seed=9).to(dev)
    with torch.no_grad():
        tgt = f.eval(src, return_names=["dino_feats"])["dino_feats"]
    t = torch.zeros(3, device=dev, requires_grad=True)
    opt = torch.optim.Adam([t], lr=1e-4)

    def it():
        opt.zero_grad()
        out = f.eval(src + t, return_names=["dino_feats"])
        loss = torch.norm(out["dino_feats"] - tgt, dim=1).mean() + 100 * torch.relu(out["dist"]).mean()
        loss.backward()
        opt.step()

    for _ in range(10):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        it()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    with torch.no_grad():
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(100):
            f.eval(src, return_names=["dino_feats"])
        torch.cuda.synchronize(); de = (time.perf_counter() - t1) / 100
    print("N=%d: optimiser iteration (eval+loss+backward+Adam) %.1f us, forward-only eval %.1f us" % (n, dt * 1e6, de * 1e6), flush=True)
