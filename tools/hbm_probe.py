#!/usr/bin/env python3
"""What the HBM of this box sustains for pure writes, pure reads and copies (torch ops on 2-GiB bf16 tensors, hipGraph-free,
HIP events): the practical ceiling of the write-dominated 1x1 expand convolutions (profiles/r05_experiments.txt)."""
import torch

dev = torch.device("cuda:0")
n = 1 << 30  # elements (bf16: 2 GiB)
a = torch.empty(n, dtype=torch.bfloat16, device=dev)
b = torch.empty(n, dtype=torch.bfloat16, device=dev)
a.fill_(1.0); b.fill_(2.0)


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 2 / 1e9
s = t(lambda: a.fill_(3.0)); print("fill   (write only)      %.2f TB/s" % (gb / s / 1e3))
s = t(lambda: a.zero_()); print("zero   (memset)          %.2f TB/s" % (gb / s / 1e3))
s = t(lambda: b.copy_(a)); print("copy   (1 read + 1 write) %.2f TB/s total, %.2f TB/s of writes" % (2 * gb / s / 1e3, gb / s / 1e3))
s = t(lambda: torch.add(a, b, out=b)); print("add    (2 reads + 1 write) %.2f TB/s total, %.2f TB/s of writes" % (3 * gb / s / 1e3, gb / s / 1e3))
af = a.view(torch.float32)
s = t(lambda: af.sum()); print("sum    (read only)       %.2f TB/s" % (gb / s / 1e3))
s = t(lambda: torch.relu_(a)); print("relu_  (1 read + 1 write, in place) %.2f TB/s total" % (2 * gb / s / 1e3))
