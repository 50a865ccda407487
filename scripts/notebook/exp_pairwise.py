"""GPU measurement: the pairwise descriptor kernel of C5 (100 000 x 300 x 384) through the shim."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3fields_amd import corr_utils as cu

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for B1, B2, C in ((100000, 300, 384), (100000, 320, 384), (100000, 1024, 384), (20000, 300, 1024)):
    src = torch.randn(B1, C, generator=g).to(dev)
    tgt = torch.randn(B2, C, generator=g).to(dev)
    for name, fn in (("dist", lambda: cu._pairwise(src, tgt, 1.0, "l2", 0, False)),
                     ("softmax+argmax", lambda: cu.nearest_descriptor(src, tgt, 1.0))):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):          # back to back: the host runs ahead, so this is device time
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 20
        print("%6d x %4d x %4d %-15s %.3f ms  (%.1f TFLOP/s at 3 flop per pair-channel)" % (B1, B2, C, name, t, 3.0 * B1 * B2 * C / t / 1e9), flush=True)
