"""GPU parity tests of the fused sample-and-contract kernels (btx_contract_fwd), all through the C-ABI.

Tolerances (rel-L2 over the output tensor):
  f32 MFMA path  vs reference outputs / f64-accumulating oracle : 1e-5   (north_star bar: 1e-4)
  bf16 MFMA path vs the oracle run on bf16-rounded operands      : 5e-4   (f32 vs f64 accumulation + a few weights
                                                                           whose bf16 rounding flips on the ~1e-6
                                                                           difference of the fast transcendentals)
  bf16 MFMA path vs the f32 reference                             : 1e-2   (8-bit mantissas; stated, not 1e-4)
  split-bf16 path (bf16x3) vs reference outputs / the oracle      : 3e-5   (hi + lo bf16 operands, three MFMAs per
                                                                           product: 16 mantissa bits; north_star bar 1e-4)
"""
import warnings

import numpy as np
import pytest
import torch

from helpers import case_geometry, oracle_forward, rel_l2

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")

TOL_F32, TOL_BF16_ORACLE, TOL_BF16_REF, TOL_X3 = 1e-5, 5e-4, 1e-2, 3e-5
TOL = {"f32": TOL_F32, "bf16": TOL_BF16_REF, "bf16x3": TOL_X3}


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _layer_from_meta(meta, dev):
    from bayesian_torch_amd import layers as L
    torch.manual_seed(meta["init_seed"])
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["kwargs"].items()}
    return getattr(L, meta["cls"])(**kw).to(dev)


def _noise_from(d, dev):
    nz = {"eps_w": torch.from_numpy(d["eps_w"]).to(dev)}
    if "eps_b" in d:
        nz["eps_b"] = torch.from_numpy(d["eps_b"]).to(dev)
    if "sign_in" in d:
        nz["sign_in"] = torch.from_numpy(d["sign_in"]).to(dev)
        nz["sign_out"] = torch.from_numpy(d["sign_out"]).to(dev)
    return nz


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("prec", ["f32", "bf16", "bf16x3"])
def test_explicit_noise_vs_reference_outputs(golden, prec, gather):
    """the reference's own noise + parameters + inputs -> must reproduce the reference's own outputs.
    gather=False: the kernel the library picks for the shape (tap-unrolled patch / patch / LDS-DMA / register-staged /
    gather); gather=True: the element-wise gather kernel on every shape (BTX_FLAG_GATHER)."""
    dev = _dev()
    for name, (meta, d) in golden["cases"].items():
        layer = _layer_from_meta(meta, dev)
        layer.precision = prec
        x = torch.from_numpy(d["x"]).to(dev)
        with torch.no_grad():
            out = layer._forward_hip(x, noise=_noise_from(d, dev), sample_idx=0, gather=gather).float().cpu().numpy()
        assert out.shape == d["out"].shape, name
        err = rel_l2(out, d["out"])
        assert err < TOL[prec], (name, prec, gather, err)
        if prec == "bf16":
            geo = case_geometry(meta)
            ob = oracle_forward(geo, d["x"], d["mu_w"], d["rho_w"], d.get("mu_b"), d.get("rho_b"), d["eps_w"],
                                d.get("eps_b"), d.get("sign_in"), d.get("sign_out"), bf16=True)
            assert rel_l2(out, ob) < TOL_BF16_ORACLE, (name, gather, rel_l2(out, ob))


# (class, kwargs, x shape) — channel counts that take the FAST granule kernels (C/groups % 8 == 0)
FUSED_CASES = [
    ("Conv2dFlipout", dict(in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=False), (2, 64, 14, 14)),
    ("Conv2dFlipout", dict(in_channels=32, out_channels=80, kernel_size=3, stride=2, padding=1), (3, 32, 15, 13)),
    ("Conv2dFlipout", dict(in_channels=64, out_channels=128, kernel_size=1, stride=2, padding=0, bias=False), (2, 64, 10, 10)),
    ("Conv2dFlipout", dict(in_channels=32, out_channels=48, kernel_size=3, padding=2, dilation=2, groups=2), (2, 32, 9, 11)),
    ("Conv2dFlipout", dict(in_channels=256, out_channels=64, kernel_size=3, padding=1, bias=False), (2, 256, 7, 7)),  # split-K
    ("Conv2dFlipout", dict(in_channels=24, out_channels=40, kernel_size=3, padding=1), (1, 24, 20, 20)),  # C%32 != 0
    ("Conv2dFlipout", dict(in_channels=64, out_channels=96, kernel_size=3, padding=1, groups=2), (3, 64, 17, 19)),  # DMA + groups
    ("Conv2dFlipout", dict(in_channels=128, out_channels=64, kernel_size=3, stride=2, padding=1), (5, 128, 23, 21)),  # DMA, M tail
    ("Conv3dReparameterization", dict(in_channels=32, out_channels=32, kernel_size=3, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, padding=1), (1, 32, 6, 7, 8)),
    ("Conv2dReparameterization", dict(in_channels=64, out_channels=96, kernel_size=3, stride=1, padding=1), (2, 64, 12, 12)),
    ("Conv2dReparameterization", dict(in_channels=128, out_channels=32, kernel_size=5, stride=2, padding=2, bias=False), (1, 128, 17, 17)),
    ("Conv1dFlipout", dict(in_channels=16, out_channels=32, kernel_size=5, stride=2, padding=2), (3, 16, 301)),
    ("Conv1dReparameterization", dict(in_channels=8, out_channels=8, kernel_size=3, padding=1), (2, 8, 50)),
    ("Conv3dFlipout", dict(in_channels=8, out_channels=16, kernel_size=3, stride=(1, 2, 1), padding=1), (2, 8, 5, 8, 6)),
    ("ConvTranspose2dFlipout", dict(in_channels=16, out_channels=16, kernel_size=4, stride=2, padding=1), (2, 16, 6, 7)),
    ("ConvTranspose2dReparameterization", dict(in_channels=16, out_channels=24, kernel_size=3, stride=2, padding=1, output_padding=1, groups=2), (2, 16, 5, 5)),
    ("LinearFlipout", dict(in_features=784, out_features=512), (256, 784)),
    ("LinearFlipout", dict(in_features=512, out_features=10), (256, 512)),
    ("LinearReparameterization", dict(in_features=128, out_features=64), (32, 128)),
    ("LinearFlipout", dict(in_features=512, out_features=1000), (64, 512)),
    # patch kernel (stride-1 2-D): several row tiles per image with a ragged last tile; several images per tile with a
    # ragged last group; asymmetric taps, dilation, no padding, groups, split over channel blocks
    ("Conv2dFlipout", dict(in_channels=64, out_channels=64, kernel_size=3, padding=1, bias=False), (2, 64, 56, 56)),
    ("Conv2dFlipout", dict(in_channels=32, out_channels=64, kernel_size=3, padding=1), (3, 32, 37, 41)),
    ("Conv2dFlipout", dict(in_channels=32, out_channels=32, kernel_size=3, padding=1), (5, 32, 14, 14)),
    ("Conv2dFlipout", dict(in_channels=512, out_channels=128, kernel_size=3, padding=1, bias=False), (13, 512, 7, 7)),
    ("Conv2dFlipout", dict(in_channels=32, out_channels=40, kernel_size=(1, 3), padding=(0, 1)), (2, 32, 9, 30)),
    ("Conv2dReparameterization", dict(in_channels=64, out_channels=32, kernel_size=(3, 1), padding=0), (2, 64, 19, 23)),
    ("Conv2dFlipout", dict(in_channels=32, out_channels=64, kernel_size=3, padding=0, dilation=(2, 3)), (2, 32, 25, 31)),
    ("Conv2dReparameterization", dict(in_channels=64, out_channels=64, kernel_size=5, padding=2, groups=2), (2, 64, 30, 30)),
    ("Conv2dFlipout", dict(in_channels=32, out_channels=16, kernel_size=3, padding=1), (1, 32, 3, 200)),
    # phase-plane kernel (3x3 / stride 2 / pad 1): row tiles with a ragged last tile, whole images, several images per tile
    # with a ragged last group, odd extents, groups, several channel blocks, split-K
    ("Conv2dFlipout", dict(in_channels=64, out_channels=128, kernel_size=3, stride=2, padding=1, bias=False), (3, 64, 56, 56)),
    ("Conv2dFlipout", dict(in_channels=128, out_channels=64, kernel_size=3, stride=2, padding=1), (2, 128, 28, 28)),
    ("Conv2dFlipout", dict(in_channels=256, out_channels=64, kernel_size=3, stride=2, padding=1, bias=False), (9, 256, 14, 14)),
    ("Conv2dFlipout", dict(in_channels=32, out_channels=64, kernel_size=3, stride=2, padding=1), (2, 32, 51, 37)),
    ("Conv2dReparameterization", dict(in_channels=64, out_channels=64, kernel_size=3, stride=2, padding=1, groups=2), (5, 64, 13, 18)),
    ("Conv2dFlipout", dict(in_channels=32, out_channels=32, kernel_size=3, stride=2, padding=1), (1, 32, 2, 2)),
    # pointwise Flipout-GEMM (btx_contract_pw.h: 1x1 / stride 1 / no padding and Linear with N % 64 == 0): resident and
    # streamed activation stages, several n-tiles per workgroup, n-tile chunks, ragged pixel tiles, bias, groups
    ("Conv2dFlipout", dict(in_channels=64, out_channels=256, kernel_size=1, bias=False), (3, 64, 19, 17)),
    ("Conv2dFlipout", dict(in_channels=256, out_channels=64, kernel_size=1), (2, 256, 14, 14)),
    ("Conv2dFlipout", dict(in_channels=128, out_channels=512, kernel_size=1, bias=False), (2, 128, 28, 28)),
    ("Conv2dFlipout", dict(in_channels=1024, out_channels=256, kernel_size=1, bias=False), (1, 1024, 14, 14)),
    ("Conv2dReparameterization", dict(in_channels=96, out_channels=128, kernel_size=1), (2, 96, 9, 31)),
    ("Conv2dFlipout", dict(in_channels=64, out_channels=128, kernel_size=1, groups=2), (2, 64, 16, 16)),
    ("LinearFlipout", dict(in_features=512, out_features=512), (256, 512)),
    ("LinearReparameterization", dict(in_features=32, out_features=64, bias=False), (300, 32)),
    ("Conv3dFlipout", dict(in_channels=32, out_channels=64, kernel_size=1), (1, 32, 3, 9, 10)),
    # 8-wave pointwise GEMM (btx_contract_gemm8.h; bf16 activations + MFMA, K >= 128, N/groups % 128 == 0): ragged pixel tile,
    # bias, groups, several n-tile pairs, odd stage counts
    ("Conv2dFlipout", dict(in_channels=256, out_channels=256, kernel_size=1, groups=2), (3, 256, 11, 13)),
    ("Conv2dFlipout", dict(in_channels=160, out_channels=384, kernel_size=1, bias=False), (2, 160, 17, 9)),
    ("LinearFlipout", dict(in_features=2048, out_features=128), (70, 2048)),
    ("Conv2dFlipout", dict(in_channels=128, out_channels=256, kernel_size=1, stride=2, bias=False), (3, 128, 15, 13)),   # strided
    ("Conv3dFlipout", dict(in_channels=128, out_channels=128, kernel_size=1, stride=(1, 2, 3)), (2, 128, 3, 9, 10)),
    # element-wise (GEN) kernels with in-kernel noise: odd channel counts
    ("Conv2dFlipout", dict(in_channels=3, out_channels=64, kernel_size=7, stride=2, padding=3, bias=False), (2, 3, 32, 32)),
    ("LinearFlipout", dict(in_features=50, out_features=10), (7, 50)),
    # small-C stems: row-fused DMA path (bf16/bf16 with even stride, f32/f32 any stride), channel padding otherwise
    ("Conv2dFlipout", dict(in_channels=3, out_channels=32, kernel_size=3, stride=2, padding=1), (3, 3, 21, 18)),
    ("Conv2dFlipout", dict(in_channels=1, out_channels=16, kernel_size=5, stride=1, padding=2), (2, 1, 12, 12)),
    ("Conv2dReparameterization", dict(in_channels=4, out_channels=64, kernel_size=(5, 7), stride=(1, 2), padding=(2, 3)), (2, 4, 9, 30)),
    ("Conv2dReparameterization", dict(in_channels=5, out_channels=7, kernel_size=3, padding=1), (2, 5, 9, 9)),
]


def _run_fused(cls, kw, xshape, prec, act, dev, sample=3, seed_init=11):
    from bayesian_torch_amd import layers as L
    import bayesian_torch_amd as bt
    bt.manual_seed(2024)
    torch.manual_seed(seed_init)
    layer = getattr(L, cls)(**kw).to(dev)
    layer.precision = prec
    x = torch.randn(*xshape).to(dev)
    if act == "bf16":
        x = x.to(torch.bfloat16)
    with torch.no_grad():
        out = layer._forward_hip(x, sample_idx=sample)
        nz = layer.materialize_noise(sample, tuple(x.shape), tuple(out.shape), x.dtype)
    wn = "weight" if cls.startswith("Linear") else "kernel"
    geo = case_geometry(dict(cls=cls, kwargs=kw))
    args = dict(x=x.float().cpu().numpy(), mu_w=getattr(layer, "mu_" + wn).detach().cpu().numpy(),
                rho_w=getattr(layer, "rho_" + wn).detach().cpu().numpy(),
                mu_b=None if layer.mu_bias is None else layer.mu_bias.detach().cpu().numpy(),
                rho_b=None if layer.rho_bias is None else layer.rho_bias.detach().cpu().numpy(),
                eps_w=nz["eps_w"].cpu().numpy(), eps_b=nz["eps_b"].cpu().numpy() if "eps_b" in nz else None,
                sign_in=nz["sign_in"].cpu().numpy() if "sign_in" in nz else None,
                sign_out=nz["sign_out"].cpu().numpy() if "sign_out" in nz else None)
    return layer, x, out, geo, args


@pytest.mark.parametrize("prec,act", [("f32", "f32"), ("bf16", "f32"), ("bf16", "bf16"), ("f32", "bf16"), ("bf16x3", "f32")])
def test_fused_noise_kernels_vs_oracle(prec, act):
    """in-kernel Philox / sign hash: the oracle regenerates nothing itself here — it is fed the noise the RNG
    kernels materialise (pinned to the CPU restatement in test_gpu_rng_kl.py) and must reproduce the output"""
    dev = _dev()
    worst = 0.0
    for cls, kw, xshape in FUSED_CASES:
        layer, x, out, geo, a = _run_fused(cls, kw, xshape, prec, act, dev)
        o = out.float().cpu().numpy()
        assert np.isfinite(o).all(), (cls, kw)
        ref = oracle_forward(geo, a["x"], a["mu_w"], a["rho_w"], a["mu_b"], a["rho_b"], a["eps_w"], a["eps_b"],
                             a["sign_in"], a["sign_out"], bf16=(prec == "bf16"))
        if act == "bf16":  # the kernel rounds its f32 result to bf16 on store
            ref = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
            tol = 3e-3
        else:
            tol = {"f32": TOL_F32, "bf16": TOL_BF16_ORACLE, "bf16x3": TOL_X3}[prec]
        err = rel_l2(o, ref)
        worst = max(worst, err)
        assert err < tol, (cls, kw, xshape, prec, act, err)
        if prec == "bf16" and act == "f32":
            ref32 = oracle_forward(geo, a["x"], a["mu_w"], a["rho_w"], a["mu_b"], a["rho_b"], a["eps_w"], a["eps_b"],
                                   a["sign_in"], a["sign_out"], bf16=False)
            assert rel_l2(o, ref32) < TOL_BF16_REF
    print("worst rel-L2 (%s/%s): %.3g" % (prec, act, worst))


PRECS = ["f32", "bf16", "bf16x3"]


def _random_conv_cases(n, seed):
    """seeded random 2-D geometries that steer through every kernel variant: stems (C <= 4), channel padding,
    stride-1 patch tiles with ragged rows / several images per tile, strided and dilated LDS-DMA tiles, groups,
    ragged output-channel tiles, split-K"""
    import random
    rng = random.Random(seed)
    cases = []
    while len(cases) < n:
        cin = rng.choice([1, 3, 4, 8, 16, 24, 32, 40, 64, 96, 128])
        groups = rng.choice([1, 1, 1, 2, 4])
        if cin % groups or (cin // groups) % 8 and groups > 1:
            groups = 1
        cout = rng.choice([groups * rng.randint(1, 40), 64, 72, 128, 8 * rng.randint(1, 20)])
        cout = max(groups, cout // groups * groups)
        kh, kw = rng.choice([(1, 1), (3, 3), (3, 3), (2, 2), (5, 5), (1, 3), (3, 1), (7, 7)])
        stride = rng.choice([1, 1, 1, 2, 2, 3])
        dil = rng.choice([1, 1, 1, 2])
        pad = rng.randint(0, 3)
        h, w = rng.randint(4, 40), rng.randint(4, 40)
        if (h + 2 * pad - dil * (kh - 1) - 1) < 0 or (w + 2 * pad - dil * (kw - 1) - 1) < 0:
            continue
        nb = rng.randint(1, 9)
        typ = rng.choice(["Flipout", "Flipout", "Reparameterization"])
        kwargs = dict(in_channels=cin, out_channels=cout, kernel_size=(kh, kw), stride=stride, padding=pad,
                      dilation=dil, groups=groups, bias=rng.random() < 0.5)
        if typ == "Reparameterization":
            kwargs.update(prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0)
        cases.append(("Conv2d" + typ, kwargs, (nb, cin, h, w)))
    return cases


@pytest.mark.parametrize("prec", ["f32", "bf16", "bf16x3"])
def test_random_geometries_match_oracle(prec):
    """48 seeded random conv geometries per precision, in-kernel noise, vs the CPU oracle fed with the same noise"""
    dev = _dev()
    worst = 0.0
    for i, (cls, kw, xshape) in enumerate(_random_conv_cases(48, 20260925 + PRECS.index(prec))):
        layer, x, out, geo, a = _run_fused(cls, kw, xshape, prec, "bf16" if prec == "bf16" else "f32", dev, sample=i,
                                           seed_init=100 + i)
        o = out.float().cpu().numpy()
        assert np.isfinite(o).all(), (cls, kw, xshape)
        ref = oracle_forward(geo, a["x"], a["mu_w"], a["rho_w"], a["mu_b"], a["rho_b"], a["eps_w"], a["eps_b"],
                             a["sign_in"], a["sign_out"], bf16=(prec == "bf16"))
        if prec == "bf16":
            ref = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
        tol = {"f32": TOL_F32, "bf16": 3e-3, "bf16x3": TOL_X3}[prec]
        err = rel_l2(o, ref)
        worst = max(worst, err)
        assert err < tol, (i, cls, kw, xshape, prec, err)
    print("worst rel-L2 over random geometries (%s): %.3g" % (prec, worst))


def test_determinism_and_sample_dependence():
    dev = _dev()
    from bayesian_torch_amd import layers as L
    torch.manual_seed(0)
    layer = L.Conv2dFlipout(64, 64, 3, padding=1, bias=False).to(dev)
    x = torch.randn(4, 64, 20, 20, device=dev)
    with torch.no_grad():
        a = layer._forward_hip(x, sample_idx=5)
        b = layer._forward_hip(x, sample_idx=5)
        c = layer._forward_hip(x, sample_idx=6)
        y1, y2 = layer(x), layer(x)  # auto-incrementing sample counter
    assert torch.equal(a, b)
    assert not torch.equal(a, c) and not torch.equal(y1[0], y2[0])
    # homogeneity: every rounding commutes with a power-of-two scale -> bit-exact
    with torch.no_grad():
        d = layer._forward_hip(2.0 * x, sample_idx=5)
    assert torch.equal(d, 2.0 * a)


def test_memory_formats_and_views():
    """NCHW-contiguous, channels_last and sliced inputs give the same result"""
    dev = _dev()
    from bayesian_torch_amd import layers as L
    torch.manual_seed(0)
    layer = L.Conv2dReparameterization(32, 32, 3, padding=1).to(dev)
    x = torch.randn(2, 32, 9, 9, device=dev)
    big = torch.randn(2, 40, 9, 9, device=dev)
    big[:, 4:36] = x
    with torch.no_grad():
        a = layer._forward_hip(x, sample_idx=1)
        b = layer._forward_hip(x.contiguous(memory_format=torch.channels_last), sample_idx=1)
        c = layer._forward_hip(big[:, 4:36], sample_idx=1)
    assert a.shape == (2, 32, 9, 9) and torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("prec", ["f32", "bf16", "bf16x3"])
def test_full_size_resnet18_shapes_properties(prec):
    """BASELINE sizes (bs 64): size-independent properties instead of the (slow) CPU oracle.
    sigma -> 0 turns Flipout into the deterministic convolution: compare with torch's own f32 conv on the GPU;
    plus determinism and exact homogeneity."""
    dev = _dev()
    from bayesian_torch_amd import layers as L
    import torch.nn.functional as F
    shapes = [(64, 64, 56, 1, 3), (64, 128, 56, 2, 3), (64, 128, 56, 2, 1), (128, 128, 28, 1, 3), (256, 256, 14, 1, 3),
              (256, 512, 14, 2, 3), (512, 512, 7, 1, 3)]
    for cin, cout, hw, stride, k in shapes:
        torch.manual_seed(1)
        layer = L.Conv2dFlipout(cin, cout, k, stride=stride, padding=k // 2, bias=False).to(dev)
        layer.precision = prec
        x = torch.randn(64, cin, hw, hw, device=dev)
        with torch.no_grad():
            y = layer._forward_hip(x, sample_idx=0)
            assert torch.equal(y, layer._forward_hip(x, sample_idx=0))
            assert torch.equal(layer._forward_hip(4.0 * x, sample_idx=0), 4.0 * y)
            layer.rho_kernel.data.fill_(-100.0)  # sigma = 3.7e-44 -> the perturbation vanishes
            y0 = layer._forward_hip(x, sample_idx=0)
            ref = F.conv2d(x, layer.mu_kernel, None, stride, k // 2)
        err = float((y0 - ref).norm() / ref.norm())
        assert err < {"f32": 1e-5, "bf16": 6e-3, "bf16x3": 3e-5}[prec], (cin, cout, hw, stride, k, prec, err)
        assert float((y - y0).norm() / y0.norm()) > 1e-2  # and with sigma > 0 it really is perturbed
