"""GPU: BTX-RNG v1 kernels vs the CPU restatement, and the KL reduction vs the reference's known answers."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def test_fill_eps_matches_cpu_restatement():
    from bayesian_torch_amd import functional as BF
    from oracle import bt_oracle as o
    dev = _dev()
    for (n, seed, sample, layer, stream) in ((100003, 1234, 0, 1, 0), (4096, 2 ** 40 + 17, 7, 3, 1), (5, 9, 123456, 99, 0)):
        g = BF.fill_eps_hip(n, dev, seed, sample, layer, stream).cpu().numpy()
        c = o.eps(n, seed, sample, layer, stream)
        assert np.isfinite(g).all()
        assert np.abs(g - c).max() < 2e-5, (n, np.abs(g - c).max())


def test_fill_sign_is_bit_exact():
    from bayesian_torch_amd import functional as BF
    from oracle import bt_oracle as o
    dev = _dev()
    for (n, seed, sample, layer, stream) in ((100003, 1234, 0, 1, 2), (64 * 97, 5, 9, 2, 3)):
        g = BF.fill_sign_hip(n, dev, seed, sample, layer, stream).cpu().numpy()
        assert np.array_equal(g, o.sign(n, seed, sample, layer, stream))


def test_layer_kl_matches_reference_values(golden):
    from bayesian_torch_amd import layers as L
    dev = _dev()
    for name, (meta, d) in golden["cases"].items():
        torch.manual_seed(meta["init_seed"])
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["kwargs"].items()}
        layer = getattr(L, meta["cls"])(**kw).to(dev)
        with torch.no_grad():
            kl = float(layer.kl_loss())
            kl2 = float(layer.kl_loss())  # cached path
        assert abs(kl - meta["kl"]) <= 2e-6 * meta["kl"], (name, kl, meta["kl"])
        assert kl2 == kl


def test_model_kl_known_answers(golden):
    import bayesian_torch_amd as bt
    from bayesian_torch_amd.models.resnet import resnet18, resnet50
    dev = _dev()
    km = golden["kat"]["models"]
    base = dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, moped_delta=0.5)
    for key, arch, typ, moped in (("resnet18_Flipout", resnet18, "Flipout", False),
                                  ("resnet18_Flipout_moped", resnet18, "Flipout", True),
                                  ("resnet50_Flipout", resnet50, "Flipout", False)):
        torch.manual_seed(0)
        m = arch()
        bt.dnn_to_bnn(m, dict(base, type=typ, moped_enable=moped))
        m = m.to(dev)
        with torch.no_grad():
            kl = float(bt.get_kl_loss(m))
        assert abs(kl - km[key]["kl"]) <= 1e-5 * km[key]["kl"], (key, kl, km[key]["kl"])  # KL rel-err bar 1e-4


def test_kl_tensor_priors_and_odd_sizes():
    from bayesian_torch_amd import functional as BF
    from oracle import bt_oracle as o
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    for n in (1, 3, 1000, 4097, 262147):
        mu = torch.randn(n, generator=g) * 0.1
        rho = torch.randn(n, generator=g) * 0.5 - 3
        pm = torch.randn(n, generator=g) * 0.05
        ps = torch.rand(n, generator=g) + 0.5
        a = float(BF.kl_hip(mu.to(dev), rho.to(dev), 0.1, 0.7))
        b = o.kl_mean(mu.numpy(), rho.numpy(), 0.1, 0.7)
        assert abs(a - b) <= 2e-6 * abs(b), (n, a, b)
        a = float(BF.kl_hip(mu.to(dev), rho.to(dev), 0.0, 1.0, pm.to(dev), ps.to(dev)))
        b = o.kl_mean(mu.numpy(), rho.numpy(), 0.0, 1.0, pm.numpy(), ps.numpy())
        assert abs(a - b) <= 2e-6 * abs(b), (n, a, b)
