"""End-to-end use of the drop-in `Fusion` on synthetic data (needs an MI355X; no reference code, no network).

The flow of the reference's vis_tracking.py with the 2-D producers replaced by synthetic maps:

    frame 0 : set the observation -> select_features_rand (grid shell + mask filter + FPS + descriptors)
    frame t : new camera poses (the whole scene moves rigidly) -> the rigid_tracking optimiser (100 Adam steps through
              the HIP field query and its backward, replayed as one HIP graph) -> tracked keypoints

The scene moves by a known rigid motion per frame, so the tracking error can be printed.  `Fusion.rigid_tracking`
is the same loop with the reference's constants (lr = 0.01, fusion.py:1613); its 1-cm first steps are tuned for the
reference's real scenes and push keypoints of this flat synthetic patch below the surface (where the reference's
loss has no gradient), so the example calls the optimiser with lr = 0.003.

    python examples/track_synthetic.py [--frames 5] [--eager]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3fields_amd import Fusion, rigid, synth     # noqa: E402


def smooth_features(V, fh, fw, C, seed=71):
    g = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(fh) / fh, np.arange(fw) / fw, indexing="ij")
    out = np.zeros((V, fh, fw, C), np.float32)
    for v in range(V):
        for c in range(C):
            a, b = g.uniform(0.5, 2.5, 2) * g.choice([-1, 1], 2)
            out[v, :, :, c] = np.sin(2 * np.pi * (a * xx + b * yy) + g.uniform(0, 2 * np.pi))
    return torch.from_numpy(out)


def world_motion(t):
    """4x4 rigid motion of the whole scene at frame t (a slow drift + yaw)."""
    a = np.deg2rad(0.8 * t)
    M = np.eye(4, dtype=np.float64)
    M[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    M[:3, 3] = [0.004 * t, -0.003 * t, 0.0]
    return M


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying a HIP graph")
    args = ap.parse_args()
    dev = "cuda:0"
    V, H, W, n_kp = 4, 240, 320, 60
    sc = synth.make_scene(V, H, W, "smooth")
    feats = smooth_features(V, H // 10, W // 10, 32)
    # one "instance" (label 1) everywhere; label 0 = background is unused
    mask = torch.nn.functional.one_hot(torch.ones(V, H, W, dtype=torch.long), 2).to(torch.float32)

    f = Fusion(num_cam=V, device=dev)
    f.use_hip_graph = not args.eager
    f.H, f.W = H, W
    f.curr_obs_torch = {"depth": sc["depth"].to(dev), "K": sc["K"].to(dev), "pose": sc["pose"].to(dev),
                        "dino_feats": feats.to(dev), "mask": mask.to(dev),
                        "consensus_mask_label": ["background", "patch"]}
    box = dict(x_lower=-0.30, x_upper=-0.12, y_lower=0.07, y_upper=0.25, z_lower=-0.02, z_upper=0.01)   # ground patch clear of the spheres
    t0 = time.perf_counter()
    src_feats, src_pts, _ = f.select_features_rand(box, n_kp, per_instance=True, res=0.004, init_idx=0)
    torch.cuda.synchronize()
    print("frame 0: %d instances x %d keypoints selected in %.1f ms" % (len(src_pts), n_kp, 1e3 * (time.perf_counter() - t0)))
    src = torch.cat(src_feats, dim=0)
    pts0 = [p.copy() for p in src_pts]
    last = [p.copy() for p in src_pts]
    pose0 = sc["pose"].numpy().astype(np.float64)
    tracker = None
    for t in range(1, args.frames + 1):
        M = world_motion(t)
        pose_t = np.stack([np.concatenate([pose0[v], [[0, 0, 0, 1]]]) @ np.linalg.inv(M) for v in range(V)])[:, :3]
        f.curr_obs_torch["pose"] = torch.from_numpy(pose_t.astype(np.float32)).to(dev)   # same images, moved world
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        start = torch.from_numpy(np.stack(last)).to(dev)
        if args.eager:
            cur, loss = rigid.track_rigid(f, src, start, use_graph=False, lr=0.003)
        else:               # one capture for the whole sequence, replayed every frame
            if tracker is None:
                tracker = rigid.RigidTracker(f, len(pts0), n_kp, lr=0.003)
            cur, loss = tracker.run(f, src, start)
        cur = cur.cpu().numpy()
        dt = time.perf_counter() - t0
        last = [cur[i * n_kp:(i + 1) * n_kp] for i in range(len(pts0))]
        truth = [(p @ M[:3, :3].T + M[:3, 3]).astype(np.float32) for p in pts0]
        err = max(float(np.abs(a - b).max()) for a, b in zip(last, truth))
        moved = max(float(np.abs(a - b).max()) for a, b in zip(pts0, truth))
        print("frame %d: 100 optimiser steps %.1f ms (%s); scene moved %.1f mm, tracking error %.2f mm, loss %.4f"
              % (t, 1e3 * dt, "eager" if args.eager else "HIP graph", 1e3 * moved, 1e3 * err, float(loss)))


if __name__ == "__main__":
    main()
