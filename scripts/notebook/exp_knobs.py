"""GPU experiment driver (experiments build only: D3F_BUILD_EXPERIMENTS=1): one workload built once, then a list of
D3F_EXP_* knob settings timed in-process -- fused kernel time from d3f_profile_next_eval, bit-identity of every output
against the first variant, and the kernel the plan reports.

    python scripts/exp_knobs.py c2_dense "tag:K1=v1,K2=v2" "tag2:K=v" ...       (tag 'base:' = no knobs)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda:0")
wl = sys.argv[1]
points = "grid"
if ":" in wl:
    wl, points = wl.split(":")
variants = []
for a in sys.argv[2:]:
    tag, _, kv = a.partition(":")
    variants.append((tag, dict(x.split("=") for x in kv.split(",") if x)))
f, pts, names, w, sc = bench.build_workload(wl, dev, 0, 1, points)
f.cache_point_order = False
f.record_plans = True
bytes_alg, _ = bench.algorithmic_bytes(w, pts.shape[0])
ref = None
seen = set()
for rep in range(int(os.environ.get("EXP_REPS", "1"))):
    for tag, env in variants:
        for k in seen:
            os.environ.pop(k, None)
        os.environ.update(env)
        seen.update(env)
        with torch.no_grad():
            fn = lambda: f.batch_eval(pts, return_names=names)
            try:
                # poison the allocator's free blocks: an output element the launch never writes must not look right
                junk = [torch.full((pts.shape[0], w["C"]), float("nan"), device=dev), torch.full((pts.shape[0],), float("nan"), device=dev),
                        torch.full((pts.shape[0],), 7, dtype=torch.uint8, device=dev)]
                del junk
                out = fn()
                torch.cuda.synchronize()
                t = bench.fused_kernel_time_ms(fn, 12, dev)
            except Exception as exc:
                print("%s | %s: FAILED %r" % (wl, tag, exc), flush=True)
                continue
        if ref is None:
            ref = {k: v.clone() for k, v in out.items()}
            same = "reference"
        else:
            bad = [k for k in ref if not torch.equal(out[k], ref[k])]
            same = "identical" if not bad else "DIFFERENT " + ",".join(
                "%s(%d pts, first %s)" % (k, int((out[k] != ref[k]).reshape(out[k].shape[0], -1).any(1).sum()),
                                          (out[k] != ref[k]).reshape(out[k].shape[0], -1).any(1).nonzero()[:6, 0].tolist()) for k in bad)
        print("%s %s | %-22s avg %.3f med %.3f min %.3f ms frac %.3f | %s | %s" % (wl, points, tag, t[0], t[1], t[2], bytes_alg / (t[0] * 1e-3) / 8e12, same,
                                                                     (f.last_plan() or {}).get("kernel")), flush=True)
