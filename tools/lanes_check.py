#!/usr/bin/env python3
"""GraphedMC with several MC samples per replay (one stream per sample inside the graph): throughput and equality with
the one-at-a-time results."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import bayesian_torch_amd as bt  # noqa: E402
from bayesian_torch_amd import mc  # noqa: E402

dev = torch.device("cuda:0")
bt.manual_seed(2024)
bt.set_precision("bf16")
model = bench.build_model("Flipout", dev, torch.bfloat16)
x = torch.randn(64, 3, 224, 224).to(dev).to(torch.bfloat16)
ref = None
for lanes in (1, 2, 3, 4, 6):
    g = mc.GraphedMC(model, x, kl=0.0, lanes=lanes)
    n = 12
    def go(base):
        for r in range(n // lanes):
            idx = [base + r * lanes + k for k in range(lanes)]
            g.run(idx[0]) if lanes == 1 else g.run_many(idx)
    go(1000)
    torch.cuda.synchronize()
    g.packed.zero_()
    t0 = time.perf_counter()
    go(0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = g.packed.clone()
    g.close()
    if ref is None:
        ref = res
    print("lanes %d: %.3f ms per MC sample = %.0f MC-samples/s ; max |stat - one-at-a-time| = %.3g" % (
        lanes, dt / n * 1e3, n / dt, float((res - ref).abs().max())))
