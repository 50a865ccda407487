"""Is the library's point order (a) a permutation, (b) the exact order of (27-bit Hilbert key of the 4-mm cell, index)?
    D3F_BUILD_EXPERIMENTS=1 python scripts/notebook/exp_order_check.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts", "notebook"))
import bench
from sim_cloud_tiles import hilbert_key
dev = torch.device("cuda", 0)
for name, env in (("xcd planes", {}), ("one table", {"D3F_EXP_ORDER_XCD": "-1"})):
    os.environ.pop("D3F_EXP_ORDER_XCD", None); os.environ.update(env)
    for wl, pts_kind in (("c2_patch", "random"), ("c5_track", "random"), ("c3_patch", "random")):
        f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1, pts_kind)
        f.cache_point_order = False
        f.batch_eval(pts, return_names=names); torch.cuda.synchronize()
        n = pts.shape[0]
        seg = (n * 4 + 255) // 256 * 256
        order = f._last_ws[2 * seg:2 * seg + 4 * n].view(torch.int32).cpu().numpy().astype(np.int64)
        perm = np.array_equal(np.sort(order), np.arange(n))
        p = pts.cpu().numpy().astype(np.float32)
        q = np.floor(p * np.float32(1.0 / np.float32(0.004))).astype(np.int64)
        key = hilbert_key(q)
        want = np.lexsort((np.arange(n), key))
        same = np.array_equal(order, want)
        mono = bool((np.diff(key[order]) >= 0).all()) if perm else None
        print("%-10s %-9s n=%8d permutation %s  keys ascending %s  equals lexsort(key, index) %s" % (name, wl, n, perm, mono, same), flush=True)
        if perm and not mono:
            ko = key[order]
            bad = np.nonzero(np.diff(ko) < 0)[0]
            print("   descents: %d; first at %s; cells (key >> 9) around the first: %s; same 18-bit cell across the descent: %.3f" % (
                len(bad), bad[:5].tolist(), (ko[bad[0] - 2:bad[0] + 4] >> 9).tolist(), float(((ko[bad] >> 9) == (ko[bad + 1] >> 9)).mean())), flush=True)
        del f, pts
        torch.cuda.empty_cache()
