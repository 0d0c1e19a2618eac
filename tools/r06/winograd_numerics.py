#!/usr/bin/env python3
"""Step 1 of the Winograd go/no-go: the bf16 error of F(2x2, 3x3) on the real stride-1 3x3 Flipout layers of
the benched ResNet18 (reference conv_flipout.py:376-417), CPU only.

For one layer per stage (its real input — the activations the converted model produces for a random 224^2 batch — its mu, rho and
one draw of eps / signs) three evaluations of   out = conv(x, mu) + s_out * conv(x * s_in, sigma * eps):
  ref      f64 direct convolution of the bf16-stored activations with the f32 weights (what the kernel's f32 mode computes);
  direct   today's bf16 path: weights rounded to bf16, products exact, f32 accumulation (emulated in f64 then compared);
  wino     F(2x2,3x3): U = G w G^T in f32 -> bf16;  V = B^T d B in f32 from the bf16 activations (twice: x and x * s_in — the sign
           does not commute with B^T . B) -> bf16;  M = sum_c U * V accumulated in f32;  Y = A^T M A in f32; s_out after that.
Prints rel-L2 against ref per layer.  Bars: per layer <= 1e-2 (tests/test_gpu_at_size.py), logits <= 1e-2.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def wino_conv(x, w):
    """x [N,C,H,W] (values already bf16-representable), w [O,C,3,3] f32 -> [N,O,H,W], pad 1; bf16 U and V, wide accumulation"""
    N, C, H, W = x.shape
    O = w.shape[0]
    U = bf16((G @ w.double() @ G.T).float())                            # [O,C,4,4], transform in f32 then rounded
    th, tw = (H + 1) // 2, (W + 1) // 2                                # odd extents (7x7): one more tile, the surplus row / column is cropped
    xp = torch.nn.functional.pad(x, (1, 1 + 2 * tw - W, 1, 1 + 2 * th - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                              # [N,C,th,tw,4,4]
    V = bf16((BT @ d.double() @ BT.T).float())                          # f32 transform of bf16 data, rounded
    M = torch.einsum("ocab,nchwab->nohwab", U, V)                       # exact products, wide accumulation (the MFMA adds in f32)
    M = M.float().double()
    Y = AT @ M @ AT.T                                                   # [N,O,th,tw,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, O, 2 * th, 2 * tw)[:, :, :H, :W]


def main():
    import bayesian_torch_amd as bt
    from bayesian_torch_amd.models import resnet
    torch.manual_seed(0)
    m = resnet.resnet18()
    bt.dnn_to_bnn(m, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type="Flipout",
                          moped_enable=False, moped_delta=0.5))
    m.eval()
    torch.manual_seed(1234)
    x0 = torch.randn(2, 3, 224, 224)
    names = ["layer1.0.conv1", "layer1.1.conv2", "layer2.0.conv2", "layer3.1.conv1", "layer4.1.conv2"]
    grabbed = {}
    mods = dict(m.named_modules())
    hs = [mods[n].register_forward_hook(lambda mod, i, o, n=n: grabbed.__setitem__(n, i[0].detach())) for n in names]
    with torch.no_grad():
        m(x0)
    for h in hs:
        h.remove()
    print("%-16s %-22s %10s %10s %8s" % ("layer", "input", "direct bf16", "F(2x2,3x3)", "ratio"))
    worst = 0.0
    for n in names:
        mod = mods[n]
        x = bf16(grabbed[n])                                            # activations as they sit in HBM
        mu = mod.mu_kernel.detach()
        sigma = torch.log1p(torch.exp(mod.rho_kernel.detach()))
        g = torch.Generator().manual_seed(7)
        delta = sigma * torch.randn(mu.shape, generator=g)
        s_in = (torch.randint(0, 2, x.shape, generator=g) * 2 - 1).double()
        Nn, O = x.shape[0], mu.shape[0]
        s_out = (torch.randint(0, 2, (Nn, O, x.shape[2], x.shape[3]), generator=g) * 2 - 1).double()
        conv = lambda a, w: torch.nn.functional.conv2d(a, w.double(), padding=1)  # noqa: E731
        ref = conv(x, mu) + s_out * conv(x * s_in, delta)
        direct = conv(x, bf16(mu)) + s_out * conv(x * s_in, bf16(delta))
        wino = wino_conv(x, mu) + s_out * wino_conv(x * s_in, delta)
        e_d = float((direct - ref).norm() / ref.norm())
        e_w = float((wino - ref).norm() / ref.norm())
        worst = max(worst, e_w)
        print("%-16s %-22s %10.3g %10.3g %8.2f" % (n, tuple(x.shape), e_d, e_w, e_w / e_d))
    print("worst F(2x2,3x3) per-layer rel-L2 %.3g (bar 1e-2)" % worst)


if __name__ == "__main__":
    main()
