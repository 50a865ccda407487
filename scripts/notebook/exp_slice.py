"""GPU experiment: would channel-sliced passes give L2 reuse on dense maps?
Times the existing kernel on a 32-channel strided VIEW of the 384-channel dense map (one 128-B
line per texel, same addresses a sliced pass would touch); 12 x that time predicts a sliced kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
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
XCD, NOR, FOR = 1 << 12, 1 << 13, 1 << 14


def timeit(names, flags, reps=8):
    f.tuning_flags = flags
    with torch.no_grad():
        for _ in range(2):
            f.batch_eval(pts, return_names=names)
        ev = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f.batch_eval(pts, return_names=names); b.record(); ev.append((a, b))
        torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for cs in (32, 64, 96, 128, 192):
    f.curr_obs_torch["slice"] = feats[..., :cs]
    f._finite_cache.clear()
    row = []
    for name, fl in [("noreorder", NOR), ("reorder", FOR), ("reorder+xcd", FOR | XCD), ("reorder+xcd t6", FOR | XCD | (6 << 8)),
                     ("reorder+xcd t5", FOR | XCD | (5 << 8)), ("reorder+xcd t8", FOR | XCD | (8 << 8)), ("reorder t6", FOR | (6 << 8)),
                     ("reorder+xcd t6 p40", FOR | XCD | (6 << 8) | (40 << 16)), ("reorder+xcd t7 p48", FOR | XCD | (7 << 8) | (48 << 16))]:
        t = timeit(["slice"], fl)
        row.append("%s: %.3f (x%d=%.2f)" % (name, t, C // cs, t * C / cs))
    print("C_slice=%d | %s" % (cs, " | ".join(row)), flush=True)
print("full C=384:", " | ".join("%s: %.3f" % (n, timeit(["dino_feats"], fl)) for n, fl in [("auto", 0), ("noreorder", NOR), ("reorder+xcd", FOR | XCD)]))
