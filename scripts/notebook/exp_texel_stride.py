"""C2-dense with the SAME 384 channels stored at a padded texel stride (a view [..., :384] of a wider buffer): what the
producer of the feature maps could buy the query by padding (the C-ABI takes any strides).  L2 background: scripts/microbench/l2_probe.hip.
    python scripts/exp_texel_stride.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "c2_dense"
f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1, "grid")
f.record_plans = True
dense = f.curr_obs_torch["dino_feats"]
V, H, W, C = dense.shape
ref = None
with torch.no_grad():
    strides = [int(x) for x in sys.argv[2:]] or [C, C + 32, C + 64, C + 96, C + 128, C + 160, 2 * C]
    for stride in strides:
        if stride == C:
            m = dense
        else:
            buf = torch.zeros(V, H, W, stride, device=dev)
            buf[..., :C] = dense
            m = buf[..., :C]
        f.curr_obs_torch["dino_feats"] = m
        fn = lambda: f.batch_eval(pts, return_names=names)
        out = fn(); torch.cuda.synchronize()
        if ref is None:
            ref = out["dino_feats"].clone()
        same = torch.equal(out["dino_feats"], ref)
        t = bench.fused_kernel_time_ms(fn, 12, dev)
        print("%s, texel stride %4d floats = %4d B (%2d lines): avg %.3f med %.3f min %.3f ms | identical %s | %s" % (
            wl, stride, stride * 4, stride * 4 // 128, t[0], t[1], t[2], same, f.last_plan()["kernel"]), flush=True)
        del m
        if stride != C:
            del buf
        torch.cuda.empty_cache()
