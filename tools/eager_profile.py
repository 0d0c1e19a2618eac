#!/usr/bin/env python3
"""Host-side cost of an eager (no hipGraph) MC step: cProfile over 30 steps of the bench model."""
import cProfile
import os
import pstats
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
packed = torch.zeros(mc.packed_numel(64, 1000), dtype=torch.float32, device=dev)
pres = "--no-presample" not in sys.argv


def step(s):
    bt.set_sample_index(model, s, presample=pres)
    mc.accumulate(packed, model(x), 0.0)


with torch.no_grad():
    for s in range(15):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(30):
        step(100 + s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host issue time per step %.3f ms, incl. final sync %.3f ms" % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for s in range(30):
        step(200 + s)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
