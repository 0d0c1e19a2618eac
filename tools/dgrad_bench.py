"""Data-gradient launches of the strided ResNet18 convolutions (batch 64, bf16 activations, Flipout): the transposed contraction
with its pixels in raster order (BTX_NO_PAR_MAJOR=1, tuning build) and parity-major (ContractParams.par_major) — time per
autograd._data_grad_hip call (operands + sampling + contraction [+ split-K reduction]) and agreement of the two.

    BTX_LIB=build_variants/libbtx_tune.so python tools/dgrad_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesian_torch_amd as bt  # noqa: E402
from bayesian_torch_amd import autograd as ag, layers as L  # noqa: E402

SHAPES = [("3x3 s2  64->128 56", 64, 128, 56, 3), ("3x3 s2 128->256 28", 128, 256, 28, 3), ("3x3 s2 256->512 14", 256, 512, 14, 3),
          ("1x1 s2  64->128 56", 64, 128, 56, 1), ("1x1 s2 128->256 28", 128, 256, 28, 1), ("1x1 s2 256->512 14", 256, 512, 14, 1)]


def run(layer, dy, x_shape, iters, env):
    os.environ.pop("BTX_NO_PAR_MAJOR", None)
    os.environ.update(env)
    mu, rho = layer._w()
    call = lambda: ag._data_grad_hip(layer, dy, x_shape, {}, 3, True, mu, rho)  # noqa: E731
    out = call()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        t0.record()
        for _ in range(iters):
            call()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) * 1e3 / iters)
    return best, out.float().clone()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    bt.manual_seed(1)
    bt.set_precision("bf16")
    print("# us per data-gradient call, batch 64, bf16, Flipout (min of 3 x %d calls)" % a.iters)
    print("%-22s %10s %12s   %s" % ("layer", "raster", "parity-major", "rel-L2 between the two"))
    for label, cin, cout, hw, k in SHAPES:
        torch.manual_seed(0)
        layer = L.Conv2dFlipout(cin, cout, k, stride=2, padding=k // 2, bias=False).to(dev)
        ho = (hw + 2 * (k // 2) - k) // 2 + 1
        dy = (torch.randn(64, cout, ho, ho, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        xs = (64, cin, hw, hw)
        tr, r = run(layer, dy, xs, a.iters, {"BTX_NO_PAR_MAJOR": "1"})
        tp, q = run(layer, dy, xs, a.iters, {})
        print("%-22s %10.1f %12.1f   %.2e" % (label, tr, tp, float((q - r).norm() / r.norm())))


if __name__ == "__main__":
    main()
