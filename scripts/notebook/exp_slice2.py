"""GPU experiment: channel slices of the dense 384-channel map through the CURRENT kernel.
Kernel-only time (d3f_profile_next_eval) of a strided [..., c0:c0+cs] view of the map; S x that
time predicts an in-kernel variant where XCD k gathers slice k % S (L2 footprint / S per XCD)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from d3fields_amd import Fusion, create_init_grid, synth

dev = torch.device("cuda:0")
V, H, W, C = 4, 480, 640, 384
sc = synth.make_scene(V, H, W, "smooth")
f = Fusion(num_cam=V, device="cuda:0")
f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
feats = synth.random_map(V, H, W, C, seed=1, device=dev)
f.curr_obs_torch["dino_feats"] = feats
f.H, f.W = H, W
pts, _ = create_init_grid(synth.WORK_BOX, 0.005)
pts = pts.to(dev)


def ktime(names, flags=0):
    f.tuning_flags = flags
    with torch.no_grad():
        fn = lambda: f.batch_eval(pts, return_names=names)
        fn(); fn()
        return bench.fused_kernel_time_ms(fn, 8, dev)[1]


print("full C=384 kernel: %.3f ms" % ktime(["dino_feats"]), flush=True)
for cs in (32, 48, 64, 96, 128, 192):
    f.curr_obs_torch["slice"] = feats[..., :cs]
    f._finite_cache.clear()
    row = []
    for name, fl in [("auto", 0), ("t4", 4 << 8), ("t5", 5 << 8), ("t6", 6 << 8), ("t4 c4", (4 << 8) | (4 << 29)), ("t5 c5", (5 << 8) | (5 << 29))]:
        t = ktime(["slice"], fl)
        row.append("%s: %.3f (x%d=%.2f)" % (name, t, C // cs, t * C / cs))
    print("C_slice=%d | %s" % (cs, " | ".join(row)), flush=True)
