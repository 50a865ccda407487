"""GPU micro-benchmark: device time of the two order probes and of the cloud ordering (HIP events around 50 calls)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3fields_amd import _lib, synth
lib = _lib.load()
dev = torch.device("cuda:0")
st = _lib.current_stream_handle(dev)
for n in (100000, 1000000):
    pts = synth.random_cloud(n, seed=3).to(dev)
    out = torch.zeros(8, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.d3f_eval_workspace_bytes(n), dtype=torch.uint8, device=dev)
    def t(fn, reps=50):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    print("n=%d locality %.1f us | lattice probe %.1f us" % (
        n, t(lambda: lib.d3f_point_order_locality(_lib.ptr(pts), n, _lib.ptr(out), st)),
        t(lambda: lib.d3f_lattice_probe(_lib.ptr(pts), n, _lib.ptr(out), st))), flush=True)
