"""The flow of the reference's vis_repr.py on synthetic data (needs an MI355X; no reference code, no network):

    update(obs)                                   colour / depth / pose / K of V views; a stand-in feature extractor
    text_queries_for_inst_mask_no_track(...)      the producer returns what Grounded-SAM returns PER VIEW (synthetic detections
                                                  with dropped, split and spurious masks); the multi-view association
                                                  (align_instance_mask_v3, fusion.py:1065-1098) runs here
    batch_eval(grid, return_names=[])             the signed-distance volume of the workspace   (vis_repr.py:88-93)
    surface points                                where the reference runs marching cubes (fusion.py:1313-1330) this takes the
                                                  grid points of the surface shell -- meshing is out of scope
    batch_eval(surface, ['dino_feats','mask','color_tensor'])      the reference's feature query   (vis_repr.py:97-103)

    python examples/repr_synthetic.py [--step 0.004]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3fields_amd import Fusion, create_init_grid, onehot2instance, synth     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step", type=float, default=0.004)
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args()
    dev = "cuda:0"
    V, H, W, C = 4, 240, 320, 384
    sc = synth.make_scene(V, H, W, "smooth")
    K, pose, depth = sc["K"].numpy(), sc["pose"].numpy(), sc["depth"].numpy()
    colour = np.random.default_rng(1).integers(0, 256, (V, H, W, 3), dtype=np.uint8)
    feats = synth.random_map(V, H // 10, W // 10, C, seed=2, device=dev)
    detections = synth.multiview_segmentation(K, pose, depth, seed=args.seed)

    def grounded_sam_stand_in(fusion, queries, thresholds, boundaries, **kw):
        gs, labels, confs = detections
        return {"mask_gs": gs, "mask_label": labels, "mask_conf": confs}

    f = Fusion(num_cam=V, device=dev, feature_extractor=lambda color, params: feats, mask_producer=grounded_sam_stand_in)
    f.update({"color": colour, "depth": depth, "pose": pose, "K": K})
    box = dict(synth.WORK_BOX)
    queries = ["mug", "box", "pen"]
    f.text_queries_for_inst_mask_no_track(queries, [0.3] * len(queries), box)        # first call: library start-up, allocations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.text_queries_for_inst_mask_no_track(queries, [0.3] * len(queries), box)
    torch.cuda.synchronize()
    n_det = [len(l) for l in detections[1]]
    print("association: %s detections per view -> %d instances %s in %.1f ms"
          % (n_det, f.get_inst_num(), f.curr_obs_torch["consensus_mask_label"], 1e3 * (time.perf_counter() - t0)))

    grid, shape = create_init_grid(box, args.step)
    grid = grid.to(dev)
    with torch.no_grad():
        f.batch_eval(grid, return_names=[])                                     # warm-up (probes, plans)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vol = f.batch_eval(grid, return_names=[])
        torch.cuda.synchronize()
        t_vol = time.perf_counter() - t0
        print("distance volume: %d x %d x %d = %d points in %.2f ms (%.2e points/s), %.1f %% valid"
              % (*shape, grid.shape[0], 1e3 * t_vol, grid.shape[0] / t_vol, 100.0 * float(vol["valid_mask"].float().mean())))
        _, surface = f.grid_shell(box, args.step, dist_threshold=args.step)
        names = ["dino_feats", "mask", "color_tensor"]
        f.record_plans = True
        f.batch_eval(surface, return_names=names)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = f.batch_eval(surface, return_names=names)
        torch.cuda.synchronize()
        t_q = time.perf_counter() - t0
    inst = onehot2instance(out["mask"])
    hist = torch.bincount(inst.to(torch.int64), minlength=f.get_inst_num()).tolist()
    print("feature query: %d surface points x (%d-d features + %d-instance mask + colour) in %.2f ms (%.2e points/s)"
          % (surface.shape[0], C, f.get_inst_num(), 1e3 * t_q, surface.shape[0] / t_q))
    print("surface points per instance:", dict(zip(["%d:%s" % (k, n) for k, n in enumerate(f.curr_obs_torch["consensus_mask_label"])], hist)))
    print("launch:", (f.last_plan() or {}).get("kernel"))
    return f, out


if __name__ == "__main__":
    main()
