import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from bayesian_torch_amd import layers as L
dev = torch.device("cuda:0")
fam = sys.argv[1] if len(sys.argv) > 1 else "Flipout"
xs = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (2, 3, 224, 224)
torch.manual_seed(11)
layer = getattr(L, "Conv2d" + fam)(in_channels=3, out_channels=64, kernel_size=7, stride=2, padding=3, bias=False).to(dev)
layer.precision = "bf16"
x = torch.randn(*xs, device=dev).to(torch.bfloat16)
scale = (torch.rand(64, device=dev) + 0.5).contiguous(); shift = torch.randn(64, device=dev).contiguous()
with torch.no_grad():
    layer._btx_sample = 5
    conv = layer.forward_fused(x, scale, shift, None, True)
    ref = F.max_pool2d(conv.float(), 3, 2, 1)
    layer._btx_sample = 5
    got = layer.forward_fused(x, scale, shift, None, True, pool=True).float()
bad = (got != ref)
print("mismatches", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()
if len(idx):
    print("images", sorted(set(idx[:, 0].tolist())))
    print("channels", sorted(set(idx[:, 1].tolist())))
    print("rows", sorted(set(idx[:, 2].tolist())))
    print("cols", sorted(set(idx[:, 3].tolist())))
    for i in idx[:10].tolist():
        n, c, r, w = i
        print(i, "got", float(got[n, c, r, w]), "ref", float(ref[n, c, r, w]), "window", conv[n, c, max(0, 2*r-1):2*r+2, max(0, 2*w-1):2*w+2].float().tolist())
