"""per-wave phase timers of stem_pool_kernel (needs BTX_LIB = a libbtx built with -DBTX_PT_TRACE -DBTX_TUNING)"""
import os, sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_torch_amd import layers as L

dev = torch.device("cuda:0")
fam = sys.argv[1] if len(sys.argv) > 1 else "Flipout"
layer = getattr(L, "Conv2d" + fam)(in_channels=3, out_channels=64, kernel_size=7, stride=2, padding=3, bias=False).to(dev)
layer.precision = "bf16"
x = torch.randn(64, 3, 224, 224, device=dev).to(torch.bfloat16)
scale = (torch.rand(64, device=dev) + 0.5).contiguous(); shift = torch.randn(64, device=dev).contiguous()
buf = torch.zeros(1 << 20, dtype=torch.int32, device=dev)
with torch.no_grad():
    for i in range(4):
        layer.forward_fused(x, scale, shift, None, True, pool=True)
    os.environ["BTX_TRACE_PTR"] = hex(buf.data_ptr())
    layer.forward_fused(x, scale, shift, None, True, pool=True)
    torch.cuda.synchronize()
    del os.environ["BTX_TRACE_PTR"]
t = buf.cpu().numpy().view(np.uint32).reshape(-1, 8)
t = t[t[:, 5] != 0]
print(fam, "waves traced:", len(t))
names = ["prologue", "sign copy", "K loops", "staging", "pool", "total", "barrier waits"]
for i, n in enumerate(names):
    c = t[:, i].astype(np.float64)
    print("  %-14s mean %9.0f  min %9.0f  max %9.0f" % (n, c.mean(), c.min(), c.max()))
for w in range(8):
    print("   wave %d: %s" % (w, t[w, :7].tolist()))
