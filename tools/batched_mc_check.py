#!/usr/bin/env python3
"""Throughput of the opt-in batched-MC mode (mc.mc_forward_batched) on the bench model, eager launches."""
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
for chunk in (1, 2, 4):
    mc.mc_forward_batched(model, x, 3 * chunk, chunk=chunk)
    torch.cuda.synchronize()
    n = 16
    t0 = time.perf_counter()
    mc.mc_forward_batched(model, x, n, chunk=chunk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("chunk %d: %.3f ms per replica = %.0f replicas/s (eager)" % (chunk, dt / n * 1e3, n / dt))
