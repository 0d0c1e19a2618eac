"""Output of one Conv2dFlipout launch (bf16, batch 256, throughput plan) saved to a file, or two such files compared:
  BTX_LIB=... python tools/libcmp.py run OUT.pt cin,cout,hw [--res]     python tools/libcmp.py cmp A.pt B.pt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    d = (a.float() - b.float())
    print("equal %s  max|d| %.3e  rel %.3e  nonzero %d of %d" % (torch.equal(a, b), float(d.abs().max()),
          float(d.norm() / b.float().norm()), int((d != 0).sum()), d.numel()))
    nz = (d != 0).nonzero()
    if len(nz):
        print("first differing (n,c,h,w):", nz[:5].tolist())
    sys.exit(0)

import bayesian_torch_amd as bt
from bayesian_torch_amd import layers as L, functional as BF
cin, cout, hw = [int(v) for v in sys.argv[3].split(",")]
dev = torch.device("cuda:0")
bt.manual_seed(3)
torch.manual_seed(0)
layer = L.Conv2dFlipout(cin, cout, 3, padding=1, bias=False).to(dev)
layer.precision = "bf16"
x = torch.randn(256, cin, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
BF._CONCURRENT = True
with torch.no_grad():
    if "--res" in sys.argv:
        res = torch.randn(256, cout, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        sc = torch.rand(cout, device=dev) + 0.5
        sh = torch.randn(cout, device=dev)
        y = layer.forward_fused(x, sc, sh, res, True)
    else:
        y = layer._forward_hip(x, sample_idx=5)
torch.save(y.cpu(), sys.argv[2])
print("saved", sys.argv[2], tuple(y.shape))
