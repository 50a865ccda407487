"""GPU experiment: effect of point order / tile size / occupancy on the fused kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3fields_amd import Fusion, create_init_grid, synth

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "dense"
V, H, W, C = 4, 480, 640, 384
fhw = (480, 640) if which == "dense" else (48, 64)
sc = synth.make_scene(V, H, W, "smooth")
f = Fusion(num_cam=V, device="cuda:0")
f.curr_obs_torch = {k: sc[k].to(dev) for k in ("depth", "K", "pose")}
f.curr_obs_torch["dino_feats"] = synth.random_map(V, fhw[0], fhw[1], C, seed=1, device=dev)
f.H, f.W = H, W
step = 0.005
pts, shape = create_init_grid(synth.WORK_BOX, step)
nx, ny, nz = shape
pts = pts.to(dev)


def part1by2(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x


idx = torch.arange(pts.shape[0], device=dev)
iz = idx % nz
iy = (idx // nz) % ny
ix = idx // (nz * ny)
morton = part1by2(ix) | (part1by2(iy) << 1) | (part1by2(iz) << 2)
perm_morton = torch.argsort(morton)


def brick_order(bx, by, bz):
    key = ((ix // bx) * ((ny + by - 1) // by) + (iy // by)) * ((nz + bz - 1) // bz) + (iz // bz)
    sub = ((ix % bx) * by + (iy % by)) * bz + (iz % bz)
    return torch.argsort(key * (bx * by * bz) + sub)


orders = {"grid(z-fast)": None, "morton": perm_morton, "brick8x8x4": brick_order(8, 8, 4),
          "brick4x4x16": brick_order(4, 4, 16), "brick16x16x55": brick_order(16, 16, 55),
          "brick40x35x55": brick_order(40, 35, 55), "random": torch.randperm(pts.shape[0], device=dev)}


def timeit(p, flags, reps=8):
    f.tuning_flags = flags
    with torch.no_grad():
        for _ in range(2):
            f.batch_eval(p, return_names=["dino_feats"])
        ev = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f.batch_eval(p, return_names=["dino_feats"]); b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def tune(tile_log2=0, xcd=False, pad_kib=0):
    return (tile_log2 << 8) | ((1 << 12) if xcd else 0) | (pad_kib << 16)


print("workload", which, "N", pts.shape[0])
NOR, FOR, NOST = 1 << 13, 1 << 14, 1 << 15
cfgs = [("auto", 0), ("no-reorder", NOR), ("reorder,nostage", NOST), ("auto again", 0)]
for name in ["grid(z-fast)", "morton", "random"]:
    perm = orders[name]
    p = pts if perm is None else pts[perm].contiguous()
    print("%-14s %s" % (name, " | ".join("%s: %.3f" % (n, timeit(p, fl)) for n, fl in cfgs)))
