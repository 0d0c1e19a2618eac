#!/usr/bin/env python3
"""bench.py's N>1 structure on one GPU: an RCCL process group (world size 1) is alive while the MC step is captured into
a hipGraph, replayed, and the packed statistics are all-reduced."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bench  # noqa: E402
import bayesian_torch_amd as bt  # noqa: E402
from bayesian_torch_amd import mc  # noqa: E402

bt.manual_seed(2024)
bt.set_precision("bf16")
model = bench.build_model("Flipout", dev, torch.bfloat16)
x = torch.randn(8, 3, 224, 224).to(dev).to(torch.bfloat16)
t = torch.ones(4, device=dev)
dist.all_reduce(t)  # communicator + watchdog up before the capture
g = mc.GraphedMC(model, x, kl=1.0, lanes=3)  # bench.py's default: three samples in flight per replay
for s in range(0, 6, 3):
    g.run_many([s, s + 1, s + 2])
dist.all_reduce(g.packed)
dist.barrier()
torch.cuda.synchronize()
u = mc.unpack(g.packed, 8, 1000)
print("samples", float(u["samples"]), "finite", bool(torch.isfinite(u["mean_prob"]).all()))
dist.destroy_process_group()
