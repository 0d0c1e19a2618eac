"""GPU: parity AT THE BASELINE.json SIZES (cfg3, cfg4, cfg5) — every variational layer of the converted model against
the reference op chain (oracle/bt_ref.py, evaluated by torch on the GPU in f32) fed with the noise BTX-RNG v1 defines
for that (layer, sample), in both precisions; the benched configuration end to end; the batched-MC mode and the
distribution of MC samples against fixtures generated from the reference itself.

Tolerances (also in DESIGN.md §2):
  f32 parity mode   per layer rel-L2 <= 1e-4 (north_star's output bar)
  bf16x3 (split-bf16, three MFMAs per product, f32 activations) per layer rel-L2 <= 1e-4 — the same bar as the f32 mode
  bf16 throughput   per layer rel-L2 <= 1e-2 against the f32 chain on the same (bf16-valued) input: operands rounded to 8
                    mantissa bits, f32 accumulation (measured 2.4-2.7e-3); logits of the graphed + fused configuration
                    <= 1e-2 for ResNet18 and ResNet50+MOPED (measured 3.8-5.5e-3)
"""
import json
import os
import warnings

import numpy as np
import pytest
import torch

BENCH_LANES = 20  # bench.py: the lanes of the driver's invocation (--steps 20)
pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")

PRIOR = dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, moped_delta=0.5)
HERE = os.path.dirname(os.path.abspath(__file__))


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _build(arch, typ, moped, dev, bf16):
    import bayesian_torch_amd as bt
    from bayesian_torch_amd.models import resnet
    torch.manual_seed(0)
    m = getattr(resnet, arch)()
    bt.dnn_to_bnn(m, dict(PRIOR, type=typ, moped_enable=moped))
    m = m.to(dev).eval()
    if bf16:
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.to(torch.bfloat16)
    bt.assign_layer_ids(m)
    return m


def _check_every_layer(model, x, typ, sample, tol):
    """forward hooks compare each variational layer with the reference chain AS IT RUNS (nothing is kept: ResNet50 at
    batch 128 moves 11 GB of activations); returns (worst rel-L2, number of layers, logits)"""
    import bayesian_torch_amd as bt
    from oracle import bt_ref
    worst = [0.0, 0]

    def hook(mod, inp, out):
        xin = inp[0].detach()
        nz = mod.materialize_noise(sample, tuple(xin.shape), tuple(out.shape), xin.dtype)
        mu, rho = mod._w()
        if mod._op.nd == 0:
            op = dict(kind="linear")
        else:
            op = dict(kind="conv", nd=2, stride=mod._op.stride[1:], padding=mod._op.padding[1:],
                      dilation=mod._op.dilation[1:], groups=mod._op.groups)
        xf = xin.float()
        if typ == "Flipout":
            ref = bt_ref.flipout_forward(xf, mu, rho, mod.mu_bias, mod.rho_bias, nz["eps_w"], nz.get("eps_b"),
                                         nz["sign_in"].float(), nz["sign_out"].float(), op)
        else:
            ref = bt_ref.reparam_forward(xf, mu, rho, mod.mu_bias, mod.rho_bias, nz["eps_w"], nz.get("eps_b"), op)
        err = float((out.detach().float() - ref).norm() / ref.norm())
        worst[0] = max(worst[0], err)
        worst[1] += 1
        assert err < tol, (mod.__class__.__name__, tuple(xin.shape), mod._op.kernel, mod._op.stride, err)
    hs = [mod.register_forward_hook(hook) for mod in model.modules() if hasattr(mod, "kl_loss")]
    try:
        with torch.no_grad():
            bt.set_sample_index(model, sample)
            logits = model(x)
    finally:
        for h in hs:
            h.remove()
    return worst[0], worst[1], logits


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("bf16", 1e-2), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("typ", ["Reparameterization", "Flipout"])
def test_resnet18_bs64_every_layer(typ, prec, tol):
    """BASELINE cfg3 (Reparameterization) / cfg4 (Flipout): dnn_to_bnn(ResNet18), 224^2, batch 64"""
    import bayesian_torch_amd as bt
    dev = _dev()
    bt.manual_seed(2024)
    bt.set_precision(prec)
    try:
        m = _build("resnet18", typ, False, dev, prec == "bf16")
        torch.manual_seed(1234)
        x = torch.randn(64, 3, 224, 224, device=dev).to(torch.bfloat16 if prec == "bf16" else torch.float32)
        worst, n, logits = _check_every_layer(m, x, typ, 5, tol)
        assert n == 21 and logits.shape == (64, 1000) and torch.isfinite(logits).all()
        print("resnet18 %s %s bs64: worst per-layer rel-L2 %.3g" % (typ, prec, worst))
    finally:
        bt.set_precision("f32")


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("bf16", 1e-2), ("bf16x3", 1e-4)])
def test_resnet50_moped_bs128_every_layer(prec, tol):
    """BASELINE cfg5 (per-GPU shard): dnn_to_bnn(ResNet50) Flipout with moped_enable=True (delta 0.5), batch 128"""
    import bayesian_torch_amd as bt
    dev = _dev()
    bt.manual_seed(2024)
    bt.set_precision(prec)
    try:
        m = _build("resnet50", "Flipout", True, dev, prec == "bf16")
        torch.manual_seed(1234)
        x = torch.randn(128, 3, 224, 224, device=dev).to(torch.bfloat16 if prec == "bf16" else torch.float32)
        worst, n, logits = _check_every_layer(m, x, "Flipout", 2, tol)
        assert n == 54 and logits.shape == (128, 1000) and torch.isfinite(logits).all()
        kat = json.load(open(os.path.join(HERE, "golden", "kat.json")))["models"].get("resnet50_Flipout_moped")
        if kat:
            with torch.no_grad():
                kl = float(bt.get_kl_loss(m))
            assert abs(kl - kat["kl"]) <= 1e-5 * abs(kat["kl"]), (kl, kat["kl"])
        print("resnet50+MOPED Flipout %s bs128: worst per-layer rel-L2 %.3g" % (prec, worst))
    finally:
        bt.set_precision("f32")


@pytest.mark.parametrize("arch,typ,moped,bs,prec,tol", [("resnet18", "Flipout", False, 64, "bf16", 1e-2),
                                                         ("resnet18", "Reparameterization", False, 64, "bf16", 1e-2),
                                                         ("resnet50", "Flipout", True, 128, "bf16", 1e-2),
                                                         ("resnet18", "Flipout", False, 64, "bf16x3", 1e-4),
                                                         ("resnet50", "Flipout", True, 128, "bf16x3", 1e-4)])
def test_benched_configuration_end_to_end(arch, typ, moped, bs, prec, tol):
    """bench.py's configurations — bf16 (headline) and bf16x3 (extra.cfg4_bf16x3 / cfg5_bf16x3: f32 activations, split-bf16
    MFMAs, logits inside north_star's 1e-4), eval-BN/ReLU/residual folded into the epilogues (fuse_resnet), one weight
    sampling launch for all lanes, hipGraph replay with BENCH_LANES MC samples as lanes of one launch per layer (bench.py's
    default) — against the UNFUSED f32-parity-mode op chain of the same parameters evaluated eagerly with the same sample
    indices (same BTX-RNG noise).  Two replays: the second one reads the cached mean tiles (BTX_SAMPLE_SKIP_MU)."""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc
    from bayesian_torch_amd.models.fuse import fuse_resnet
    dev = _dev()
    bt.manual_seed(2024)
    samples = list(range(11, 11 + 2 * BENCH_LANES))
    checked = [0, 1, BENCH_LANES - 1, BENCH_LANES, 2 * BENCH_LANES - 1]  # lanes of both replays (the f32 reference is eager: a few)
    try:
        bt.set_precision("f32")
        ref_m = _build(arch, typ, moped, dev, False)
        torch.manual_seed(1234)
        x = torch.randn(bs, 3, 224, 224, device=dev)
        refs = []
        with torch.no_grad():
            for k in checked:
                bt.set_sample_index(ref_m, samples[k])
                refs.append(ref_m(x).float().clone())
        del ref_m
        bt.set_precision(prec)
        m = _build(arch, typ, moped, dev, prec == "bf16")
        fuse_resnet(m)
        g = mc.GraphedMC(m, x.to(torch.bfloat16) if prec == "bf16" else x, kl=0.0, lanes=BENCH_LANES, keep_logits=True,
                         static_input=True)
        errs = []
        for rep in range(2):
            g.run_many(samples[rep * BENCH_LANES:(rep + 1) * BENCH_LANES])
            torch.cuda.synchronize()
            for i, k in enumerate(checked):
                if k // BENCH_LANES == rep:
                    errs.append(float((g.lane_logits[k % BENCH_LANES].float() - refs[i]).norm() / refs[i].norm()))
        g.close()
        assert len(errs) == len(checked)
        assert not torch.equal(refs[0], refs[1])
        print("%s %s%s bs%d: logits rel-L2 of the graphed %s configuration vs the unfused f32 chain: %s" % (
            arch, typ, "+MOPED" if moped else "", bs, prec, ", ".join("%.3g" % e for e in errs)))
        assert max(errs) < tol, errs
    finally:
        bt.set_precision("f32")


def test_batched_mc_matches_reference_chain():
    """mc_forward_batched = the reference's `torch.cat([data]*S)` trick.  One shared weight perturbation
    and per-example signs per chunk: its packed statistics must equal those computed from the reference op chain
    (oracle/bt_ref.py) applied layer by layer to the concatenated batch with the noise of that chunk."""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc
    from oracle import bt_ref
    dev = _dev()
    bt.manual_seed(31)
    torch.manual_seed(2)
    net = torch.nn.Sequential(torch.nn.Linear(96, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 10))
    bt.dnn_to_bnn(net, dict(PRIOR, type="Flipout", moped_enable=False))
    net = net.to(dev).eval()
    bt.assign_layer_ids(net)
    bt.set_precision("f32")
    x = torch.randn(16, 96, device=dev)
    S, chunk = 6, 4
    packed = mc.mc_forward_batched(net, x, S, chunk=chunk)
    want = torch.zeros_like(packed)
    done, cid = 0, 0
    layers = [m for m in net if hasattr(m, "kl_loss")]
    with torch.no_grad():
        while done < S:
            c = min(chunk, S - done)
            h = torch.cat([x] * c, 0)
            for i, mod in enumerate(layers):
                out_shape = (h.shape[0], mod.out_features)
                nz = mod.materialize_noise(cid, tuple(h.shape), out_shape, h.dtype)
                h = bt_ref.flipout_forward(h, mod.mu_weight, mod.rho_weight, mod.mu_bias, mod.rho_bias, nz["eps_w"],
                                           nz["eps_b"], nz["sign_in"].float(), nz["sign_out"].float(), dict(kind="linear"))
                if i + 1 < len(layers):
                    h = torch.relu(h)
            for r in range(c):
                mc.accumulate(want, h[r * 16:(r + 1) * 16].contiguous(), 0.0)
            done += c
            cid += 1
    assert torch.allclose(packed, want, rtol=1e-4, atol=1e-5), float((packed - want).abs().max())


def test_mc_distribution_matches_reference_fixture():
    """the noise streams differ from the reference's by construction, so the comparison is
    distributional: per-logit mean / variance of S = 256 MC samples of the cfg2 MLP (same init draws, same input)
    against the statistics of 256 MC samples of the REFERENCE run on torch-CPU (tests/golden/mc_stats.npz, written by
    tools/make_golden.py).  |mean - mean_ref| <= 5 * sqrt((var + var_ref) / S); variance ratio in [0.6, 1.6]."""
    import bayesian_torch_amd as bt
    dev = _dev()
    fx = np.load(os.path.join(HERE, "golden", "mc_stats.npz"))
    S, rows = int(fx["S"]), int(fx["rows"])
    bt.manual_seed(77)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(784, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(),
                              torch.nn.Linear(512, 10))
    bt.dnn_to_bnn(net, dict(PRIOR, type="Flipout", moped_enable=False))
    net = net.to(dev).eval()
    bt.assign_layer_ids(net)
    bt.set_precision("f32")
    torch.manual_seed(1234)
    x = torch.randn(256, 784).to(dev)
    ys = []
    with torch.no_grad():
        for s in range(S):
            bt.set_sample_index(net, s)
            ys.append(net(x)[:rows].float().cpu().numpy())
    y = np.stack(ys).astype(np.float64)
    mean, var = y.mean(0), y.var(0)
    mref, vref = fx["mean"].astype(np.float64), fx["var"].astype(np.float64)
    z = np.abs(mean - mref) / np.sqrt((var + vref) / S)
    ratio = var / vref
    print("MC distribution vs reference: max |z| %.2f, variance ratio %.2f..%.2f" % (z.max(), ratio.min(), ratio.max()))
    assert z.max() < 5.0, z.max()
    assert ratio.min() > 0.6 and ratio.max() < 1.6, (ratio.min(), ratio.max())


# one layer per kernel family at the BASELINE batch, compared with the reference chain evaluated ON THE CPU (ATen / MKLDNN f32):
# no GPU library sits between the HIP output and the reference arithmetic (conv_flipout.py:376-417, linear_flipout.py:149-174)
CPU_CASES = [
    # (name, class, kwargs, input shape, kernel family it lands on in bf16)
    ("taps_56", "Conv2dFlipout", dict(in_channels=64, out_channels=64, kernel_size=3, padding=1, bias=False), (64, 64, 56, 56)),
    ("taps_7", "Conv2dFlipout", dict(in_channels=512, out_channels=512, kernel_size=3, padding=1, bias=False), (64, 512, 7, 7)),
    ("taps2_s2", "Conv2dFlipout", dict(in_channels=64, out_channels=128, kernel_size=3, stride=2, padding=1, bias=False), (64, 64, 56, 56)),
    ("gemm8_pw", "Conv2dFlipout", dict(in_channels=512, out_channels=128, kernel_size=1, bias=False), (128, 512, 28, 28)),
    ("dma_k64", "Conv2dFlipout", dict(in_channels=64, out_channels=256, kernel_size=1, bias=False), (128, 64, 56, 56)),
    ("dma_n64", "Conv2dFlipout", dict(in_channels=256, out_channels=64, kernel_size=1, bias=False), (128, 256, 56, 56)),
    ("stem", "Conv2dFlipout", dict(in_channels=3, out_channels=64, kernel_size=7, stride=2, padding=3, bias=False), (64, 3, 224, 224)),
    ("fc", "LinearFlipout", dict(in_features=512, out_features=1000), (64, 512)),
    ("reparam_28", "Conv2dReparameterization", dict(in_channels=128, out_channels=128, kernel_size=3, padding=1, bias=False), (64, 128, 28, 28)),
    # 784 pixel tiles x 1 pair of n-tiles: the wide Reparameterization tile (64 px x 128 ch per wave) in a single launch, with bias
    ("reparam_wide_56", "Conv2dReparameterization", dict(in_channels=64, out_channels=128, kernel_size=3, padding=1, bias=True), (64, 64, 56, 56)),
    # stride 2, 512 pixel tiles x 1 pair of n-tiles: the wide tile of the phase-plane kernel in a single launch
    ("reparam_wide_s2", "Conv2dReparameterization", dict(in_channels=64, out_channels=128, kernel_size=3, stride=2, padding=1, bias=False), (128, 64, 56, 56)),
]


def _cpu_reference(layer, x, out_shape, sample, typ):
    from oracle import bt_ref
    nz = layer.materialize_noise(sample, tuple(x.shape), tuple(out_shape), x.dtype)
    c = lambda t: None if t is None else t.detach().float().cpu()  # noqa: E731
    mu, rho = layer._w()
    if layer._op.nd == 0:
        op = dict(kind="linear")
    else:
        op = dict(kind="conv", nd=2, stride=layer._op.stride[1:], padding=layer._op.padding[1:],
                  dilation=layer._op.dilation[1:], groups=layer._op.groups)
    mu_c = c(mu).contiguous()
    rho_c = c(rho).contiguous()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        if typ == "flipout":
            return bt_ref.flipout_forward(c(x).contiguous(), mu_c, rho_c, c(layer.mu_bias), c(layer.rho_bias), c(nz["eps_w"]).contiguous(),
                                          c(nz.get("eps_b")), c(nz["sign_in"]).contiguous(), c(nz["sign_out"]).contiguous(), op)
        return bt_ref.reparam_forward(c(x).contiguous(), mu_c, rho_c, c(layer.mu_bias), c(layer.rho_bias), c(nz["eps_w"]).contiguous(),
                                      c(nz.get("eps_b")), op)


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("bf16x3", 1e-4), ("bf16", 1e-2)])
@pytest.mark.parametrize("case", CPU_CASES, ids=[c[0] for c in CPU_CASES])
def test_kernel_families_at_baseline_batch_vs_cpu_reference(case, prec, tol):
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import layers as L
    name, cls, kw, xshape = case
    dev = _dev()
    bt.manual_seed(2024)
    torch.manual_seed(3)
    layer = getattr(L, cls)(**kw).to(dev)
    layer.precision = prec
    act = torch.bfloat16 if prec == "bf16" else torch.float32
    torch.manual_seed(1234)
    x = torch.randn(*xshape).to(dev).to(act)
    if len(xshape) == 4:
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        out = layer._forward_hip(x, sample_idx=9)
        # the launch-lanes form of the same layer (what the bench replays): lane 1 of a 2-lane launch, same sample index
        shared = kw.get("in_channels") == 3
        bt.set_sample_lanes(layer, [8, 9], batch=xshape[0])
        out2 = layer._forward_hip(x if shared else torch.cat([x, x], 0))[xshape[0]:]
        bt.set_sample_lanes(layer, None)
    torch.cuda.synchronize()
    ref = _cpu_reference(layer, x, out.shape, 9, "flipout" if "Flipout" in cls else "reparam")
    e1 = float((out.float().cpu() - ref).norm() / ref.norm())
    e2 = float((out2.float().cpu() - ref).norm() / ref.norm())
    print("%s %s batch %d vs CPU reference chain: rel-L2 %.3g (single launch) %.3g (lane of a 2-lane launch)" % (name, prec, xshape[0], e1, e2))
    assert e1 < tol and e2 < tol, (name, prec, e1, e2)


def test_stem_with_fused_bn_relu_maxpool_at_bs64_vs_cpu_reference():
    """the one-launch stem of the bench (conv1 + eval-BN + ReLU + MaxPool2d(3,2,1), bf16) at batch 64 against the CPU chain:
    reference conv_flipout.py:376-417 -> BatchNorm2d(eval) -> ReLU -> MaxPool2d, all ATen f32 on the host"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import layers as L
    dev = _dev()
    bt.manual_seed(2024)
    torch.manual_seed(3)
    layer = L.Conv2dFlipout(3, 64, 7, stride=2, padding=3, bias=False).to(dev)
    layer.precision = "bf16"
    torch.manual_seed(1234)
    x = torch.randn(64, 3, 224, 224).to(dev).to(torch.bfloat16)
    scale = (0.5 + torch.rand(64)).to(dev)
    shift = (0.1 * torch.randn(64)).to(dev)
    assert layer.pool_fusable(x)
    with torch.no_grad():
        got = layer.forward_fused(x, scale, shift, None, True, pool=True)
    torch.cuda.synchronize()
    ref = _cpu_reference(layer, x, (64, 64, 112, 112), layer._btx_sample - 1, "flipout")
    ref = torch.relu(ref * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1))
    ref = torch.nn.functional.max_pool2d(ref, 3, 2, 1)
    err = float((got.float().cpu() - ref).norm() / ref.norm())
    print("stem + BN + ReLU + maxpool bf16 batch 64 vs CPU reference chain: rel-L2 %.3g" % err)
    assert got.shape == (64, 64, 56, 56) and err < 1e-2, err
