"""GPU measurement: grid-native paths at the size of select_features_rand's 1-mm grid (~1.2e8 points)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3fields_amd import Fusion, create_init_grid, synth, fps

dev = torch.device("cuda:0")
V, H, W = 4, 480, 640
sc = synth.make_scene(V, H, W, "smooth")
f = Fusion(num_cam=V, device="cuda:0")
f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
f.curr_obs_torch["mask"] = synth.random_onehot_mask(V, H, W, 8, seed=2, device=dev)
f.curr_obs_torch["dino_feats"] = synth.random_map(V, 48, 64, 384, seed=1, device=dev)
f.H, f.W = H, W


def t_ms(fn, reps=5):
    with torch.no_grad():
        fn(); fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


for res in (0.004, 0.002, 0.001):
    grid, shape = create_init_grid(synth.WORK_BOX, res)
    n = grid.shape[0]
    g = grid.to(dev)
    a = t_ms(lambda: f.batch_eval(g, return_names=[]))
    b = t_ms(lambda: f.eval_grid(synth.WORK_BOX, res))
    c = t_ms(lambda: f.batch_eval(g, return_names=["mask"]))
    d = t_ms(lambda: f.eval_grid(synth.WORK_BOX, res, return_names=["mask"]))
    e = t_ms(lambda: f.grid_shell(synth.WORK_BOX, res, 0.005))
    idx, pts = f.grid_shell(synth.WORK_BOX, res, 0.005)
    print("res %.3f N=%d | dist-only: batch_eval %.2f ms, eval_grid %.2f ms (%.3g pts/s) | +mask: batch_eval %.2f, eval_grid %.2f | "
          "grid_shell %.2f ms -> %d survivors (%.2f%%)" % (res, n, a, b, n / b * 1e3, c, d, e, idx.numel(), 100.0 * idx.numel() / n), flush=True)
    del g, grid
    torch.cuda.empty_cache()
cloud = pts[:200000].contiguous()
print("fps 100 of %d: %.2f ms" % (cloud.shape[0], t_ms(lambda: fps(cloud, 100, init_idx=0))))
