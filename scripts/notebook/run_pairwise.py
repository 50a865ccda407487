"""A few launches of the C5 pairwise kernel (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3fields_amd import corr_utils as cu
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
src = torch.randn(100000, 384, generator=g).to(dev)
tgt = torch.randn(300, 384, generator=g).to(dev)
for _ in range(6):
    cu.nearest_descriptor(src, tgt, 1.0)
torch.cuda.synchronize()
