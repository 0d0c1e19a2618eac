"""the fused stem + max-pool launch, a few times — the target of rocprofv3 --pmc runs (tools/gpu_r3b.sh)"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_torch_amd import layers as L

dev = torch.device("cuda:0")
layer = L.Conv2dFlipout(in_channels=3, out_channels=64, kernel_size=7, stride=2, padding=3, bias=False).to(dev)
layer.precision = "bf16"
x = torch.randn(64, 3, 224, 224, device=dev).to(torch.bfloat16)
scale = (torch.rand(64, device=dev) + 0.5).contiguous()
shift = torch.randn(64, device=dev).contiguous()
with torch.no_grad():
    for i in range(6):
        layer.forward_fused(x, scale, shift, None, True, pool=True)
torch.cuda.synchronize()
