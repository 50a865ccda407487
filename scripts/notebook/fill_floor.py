"""Store-only floor of the GPU box: torch's fill kernel over a buffer the size of C2-patch's / ref-patch's output (HIP events)."""
import torch
dev = torch.device("cuda:0")
for gb in (1.51, 3.0, 7.9):
    x = torch.empty(int(gb * 1e9 / 4), dtype=torch.float32, device=dev)
    for _ in range(3): x.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): x.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("fill %.2f GB: %.3f ms = %.2f TB/s" % (gb, ms, gb / ms))
    del x
