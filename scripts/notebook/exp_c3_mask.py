"""What does the 8-channel mask cost when it rides along with the channel-sliced gather of C3-dense?
    python scripts/exp_c3_mask.py        (fused kernel time of the three launches, HIP events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
f, pts, names, w, sc = bench.build_workload("c3_dense", dev, 0, 1, "grid")
f.record_plans = True
with torch.no_grad():
    for nm in (["dino_feats", "mask"], ["dino_feats"], ["mask"], []):
        fn = lambda: f.batch_eval(pts, return_names=nm)
        fn(); torch.cuda.synchronize()
        t = bench.fused_kernel_time_ms(fn, 12, dev)
        print("c3_dense grid, return_names=%-24s avg %.3f med %.3f min %.3f ms | %s | tile %s" % (nm, t[0], t[1], t[2], f.last_plan()["kernel"], f.last_plan()["tile_points"]), flush=True)
