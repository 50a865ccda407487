"""Round 5: a longer seeded sweep of large clouds (the gated window / cell-run launches) against the CPU oracle than the test-suite runs.
    python scripts/notebook/exp_fuzz_clouds.py [first_seed] [count]"""
import collections
import os
import sys
import time

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch   # noqa: E402
import test_gpu_fuzz as T   # noqa: E402

dev = torch.device("cuda:0")
first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 40)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    c = T._cloud_case(seed)
    try:
        T._run(dev, c, 200 + seed)
        print("ok  ", seed, {k: c[k] for k in ("V", "C", "fhw", "NI", "color", "kind", "view", "cloud_n", "cloud_scale")}, flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, c, str(e)[:600], flush=True)
print("cases %d, failures %d, %.0f s" % (count, bad, time.time() - t0))
