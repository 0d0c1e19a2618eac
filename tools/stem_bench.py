"""GPU time of the ResNet stem chain (conv1 7x7/2 + bn + relu + maxpool 3/2/1), bs 64 bf16: one-launch (BtxEpilogue.pool)
vs stem launch + pool kernel.  20 calls per hipGraph."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesian_torch_amd as bt
from bayesian_torch_amd import layers as L, functional as BF

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for fam in ("Flipout", "Reparameterization"):
    torch.manual_seed(0)
    layer = getattr(L, "Conv2d" + fam)(in_channels=3, out_channels=64, kernel_size=7, stride=2, padding=3, bias=False).to(dev)
    layer.precision = "bf16"
    x = torch.randn(bs, 3, 224, 224, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    scale = (torch.rand(64, device=dev) + 0.5).contiguous()
    shift = torch.randn(64, device=dev).contiguous()

    def fused():
        layer._btx_sample = 1
        return layer.forward_fused(x, scale, shift, None, True, pool=True)

    def chain():
        layer._btx_sample = 1
        return BF.maxpool2d_hip(layer.forward_fused(x, scale, shift, None, True), 3, 2, 1)

    def conv_only():
        layer._btx_sample = 1
        return layer.forward_fused(x, scale, shift, None, True)

    for name, fn in (("one launch", fused), ("stem + pool kernel", chain), ("stem only", conv_only)):
        with torch.no_grad():
            s = torch.cuda.Stream(dev)
            with torch.cuda.stream(s):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    fn()
            g.replay(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        print("%-18s bs %d %-20s %7.1f us per call (incl. input pack + sampling pre-pass)" % (fam, bs, name, best))
