"""Drop-in surface of the Python face — CPU only.  Everything is compared with values recorded from
the reference itself by tools/make_golden.py (signatures, state_dict keys, KL known-answers, forward outputs)."""
import inspect
import warnings

import numpy as np
import pytest
import torch

import bayesian_torch_amd as bt
from bayesian_torch_amd import layers as L
from bayesian_torch_amd.models.resnet import resnet18, resnet50

warnings.filterwarnings("ignore")


def _build(meta):
    torch.manual_seed(meta["init_seed"])
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["kwargs"].items()}
    layer = getattr(L, meta["cls"])(**kw)
    x = torch.randn(*meta["x_shape"])
    return layer, x


def test_signatures_match_reference(golden):
    for name, sig in golden["kat"]["signatures"].items():
        assert str(inspect.signature(getattr(L, name).__init__)) == sig, name


def test_same_seed_same_parameters_and_state_dict(golden):
    """init_parameters draws mu_w, rho_w, mu_b, rho_b in the reference's order -> identical tensors per torch seed"""
    for name, (meta, d) in golden["cases"].items():
        layer, x = _build(meta)
        assert list(layer.state_dict().keys()) == meta["state_dict_keys"], name
        wn = "weight" if meta["cls"].startswith("Linear") else "kernel"
        assert np.array_equal(getattr(layer, "mu_" + wn).detach().numpy(), d["mu_w"]), name
        assert np.array_equal(getattr(layer, "rho_" + wn).detach().numpy(), d["rho_w"]), name
        assert np.array_equal(x.numpy(), d["x"]), name
        if "mu_b" in d:
            assert np.array_equal(layer.mu_bias.detach().numpy(), d["mu_b"]), name
        else:
            assert layer.mu_bias is None and layer.rho_bias is None and layer.eps_bias is None
        assert float(layer.kl_loss()) == meta["kl"], name


def test_cpu_forward_is_bit_exact_vs_reference(golden):
    """CPU tensors take the ATen op chain with the reference's torch-generator draw order (BASELINE config 0)"""
    for name, (meta, d) in golden["cases"].items():
        layer, x = _build(meta)
        with torch.no_grad():
            torch.manual_seed(meta["fwd_seed"])
            out, kl = layer(x)
        assert np.array_equal(out.numpy(), d["out"]), name
        assert float(kl) == meta["kl"], name
        wn = "weight" if meta["cls"].startswith("Linear") else "kernel"
        # observable side effect: eps_* holds the eps of the last forward
        assert np.array_equal(getattr(layer, "eps_" + wn).numpy(), d["eps_w"]), name
        layer.dnn_to_bnn_flag = True
        with torch.no_grad():
            torch.manual_seed(meta["fwd_seed"])
            out2 = layer(x)
        assert torch.equal(out2, out)


def test_attributes_and_errors():
    l = L.Conv2dReparameterization(8, 16, 3, stride=2, padding=1)
    assert l.posterior_mu_init == (0,) and l.posterior_rho_init == (-3.0,)  # 1-tuples, reference quirk
    assert (l.in_channels, l.out_channels, l.kernel_size, l.stride, l.padding, l.dilation, l.groups) == \
        (8, 16, 3, 2, 1, 1, 1)
    assert l.bias is True and l.quant_prepare is False and l.dnn_to_bnn_flag is False
    assert l.prior_weight_mu.shape == l.mu_kernel.shape and float(l.prior_weight_sigma[0, 0, 0, 0]) == 1.0
    f = L.LinearFlipout(10, 4, bias=False)
    assert f.posterior_mu_init == 0 and f.mu_weight.shape == (4, 10) and f.mu_bias is None
    with pytest.raises(ValueError, match="invalid in_channels size"):
        L.Conv2dReparameterization(6, 8, 3, groups=4)
    t = L.ConvTranspose2dFlipout(8, 6, 3, stride=2, output_padding=1)
    assert t.mu_kernel.shape == (8, 6, 3, 3) and t.output_padding == 1
    assert isinstance(l, L.BaseVariationalLayer_)
    # buffers are non-persistent: state_dict has only mu_* / rho_*
    assert set(l.state_dict()) == {"mu_kernel", "rho_kernel", "mu_bias", "rho_bias"}


def test_dnn_to_bnn_and_kl_known_answers(golden):
    km = golden["kat"]["models"]
    base = dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, moped_delta=0.5)
    for key, arch, typ, moped in (("resnet18_Reparameterization", resnet18, "Reparameterization", False),
                                  ("resnet18_Flipout", resnet18, "Flipout", False),
                                  ("resnet18_Flipout_moped", resnet18, "Flipout", True),
                                  ("resnet50_Flipout", resnet50, "Flipout", False)):
        torch.manual_seed(0)
        m = arch()
        bt.dnn_to_bnn(m, dict(base, type=typ, moped_enable=moped))
        bl = [(n, mod.__class__.__name__) for n, mod in m.named_modules() if hasattr(mod, "kl_loss")]
        assert len(bl) == km[key]["n_bayes_layers"]
        assert list(bl[0]) == km[key]["first"] and list(bl[-1]) == km[key]["last"]
        assert all(mod.dnn_to_bnn_flag for mod in m.modules() if hasattr(mod, "kl_loss"))
        cs = sum(float(q.double().sum()) for n, q in m.named_parameters() if ".mu_" in n or n.startswith("mu_"))
        assert abs(cs - km[key]["checksum_mu"]) < 1e-9 * max(1.0, abs(cs)), key  # same seed -> same parameters
        kl = float(bt.get_kl_loss(m))
        assert abs(kl - km[key]["kl"]) <= 1e-6 * km[key]["kl"], (key, kl)
    with pytest.raises(KeyError):
        bt.dnn_to_bnn(resnet18(), dict(base, type="Flipout"))  # moped_enable has no default in the reference
    assert bt.get_kl_loss(torch.nn.ReLU()) is None


def test_get_rho(golden):
    g = golden["kat"]["models"]["get_rho"]
    r = bt.utils.util.get_rho(torch.tensor(g["w"]), g["delta"])
    assert np.array_equal(r.numpy(), np.array(g["rho"], dtype=np.float32))


def test_converted_model_runs_on_cpu_and_returns_only_out():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Flatten(),
                            torch.nn.Linear(8 * 6 * 6, 5))
    bt.dnn_to_bnn(m, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0,
                          type="Flipout", moped_enable=True, moped_delta=0.5))
    assert m[0].__class__.__name__ == "Conv2dFlipout" and m[3].__class__.__name__ == "LinearFlipout"
    with torch.no_grad():
        y = m(torch.randn(2, 3, 6, 6))
    assert y.shape == (2, 5)
    kl = bt.get_kl_loss(m)
    assert kl.dim() == 0 and float(kl) > 0


def test_install_alias():
    import sys
    bt.install_alias("bayesian_torch_alias_for_test")
    import importlib
    mod = importlib.import_module("bayesian_torch_alias_for_test.layers")
    assert mod.Conv2dFlipout is L.Conv2dFlipout
    from bayesian_torch_alias_for_test.models.dnn_to_bnn import dnn_to_bnn  # noqa: F401
    for k in [k for k in sys.modules if k.startswith("bayesian_torch_alias_for_test")]:
        del sys.modules[k]


def test_cuda_path_fails_loudly_without_library(monkeypatch):
    """a GPU tensor may never silently take another path: with libbtx.so absent the binding raises"""
    from bayesian_torch_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "lib_path", lambda: "/nonexistent/libbtx.so")
    with pytest.raises(_lib.BtxError, match="no fallback"):
        _lib.lib()


def test_batched_mc_flipout_mode_on_cpu():
    """mc.mc_forward_batched (opt-in: replicas share the weight perturbation) through the ATen path: sample count and
    normalisation of the accumulated statistics"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc
    from bayesian_torch_amd import layers as L
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = L.LinearFlipout(12, 16)
            self.b = L.LinearFlipout(16, 5)

        def forward(self, x):
            return self.b(torch.relu(self.a(x, return_kl=False)), return_kl=False)

    net = Net().eval()
    x = torch.randn(4, 12)
    packed = mc.mc_forward_batched(net, x, 5, chunk=2, with_kl=True)
    u = mc.unpack(packed, 4, 5)
    assert abs(float(u["samples"]) - 5) < 1e-6
    assert torch.allclose(u["mean_prob"].sum(1), torch.ones(4), atol=1e-5)
    assert abs(float(u["kl"]) - float(bt.get_kl_loss(net))) < 1e-4 * abs(float(bt.get_kl_loss(net)))


# ---- round 2: advisor findings ----------------------------------------------------------------------------------
def _small_net():
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 16, 3, stride=2, bias=False), torch.nn.Flatten(),
                               torch.nn.Linear(16 * 7 * 7, 10))


@pytest.mark.parametrize("typ", ["Reparameterization", "Flipout"])
def test_moped_function_matches_reference_kl(golden, typ):
    """utils.util.MOPED(): tensor priors + posterior init from a deterministic model; KL known answer from the reference"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd.utils.util import MOPED
    kat = golden["kat"]["models"]["moped_fn_" + typ]
    torch.manual_seed(0)
    det = _small_net()
    det[1].running_mean.normal_()
    det[1].running_var.uniform_(0.5, 2.0)
    torch.manual_seed(1)
    bnn = _small_net()
    torch.manual_seed(1)
    bnn = _small_net()
    bt.dnn_to_bnn(bnn, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type=typ,
                            moped_enable=False, moped_delta=0.5))
    MOPED(bnn, _small_net(), det.state_dict(), kat["delta"])
    assert float(bt.get_kl_loss(bnn)) == pytest.approx(kat["kl"], rel=1e-6)
    assert float(bnn[1].running_mean.sum()) == pytest.approx(kat["bn_mean_sum"], rel=1e-6)
    assert torch.equal(bnn[0].mu_kernel.data, det[0].weight.data) and torch.equal(bnn[0].prior_weight_mu, det[0].weight.data)


def test_fuse_resnet_keeps_the_checkpoint_format():
    """fuse_resnet must not change state_dict keys (advisor: 144 -> 263 keys, downsample entries lost)"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd.models.resnet import resnet18
    from bayesian_torch_amd.models.fuse import fuse_resnet
    params = dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type="Flipout",
                  moped_enable=False, moped_delta=0.5)
    torch.manual_seed(0)
    a = resnet18()
    bt.dnn_to_bnn(a, params)
    keys = list(a.state_dict().keys())
    torch.manual_seed(1)
    b = resnet18()
    bt.dnn_to_bnn(b, params)
    assert fuse_resnet(b) == 9
    assert list(b.state_dict().keys()) == keys
    b.load_state_dict(a.state_dict(), strict=True)     # unfused checkpoint -> fused model
    a.load_state_dict(b.state_dict(), strict=True)     # and back
    assert isinstance(b.layer2[0].downsample, torch.nn.Sequential)
    x = torch.randn(1, 3, 224, 224)
    torch.manual_seed(5)
    ya = a.eval()(x)
    torch.manual_seed(5)
    yb = b.eval()(x)                                   # CPU: forward_fused == the same ATen chain + affine
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4 * float(ya.abs().max()))
    b.bn1.running_var.mul_(4.0)                        # folded scale/shift follow the BN tensors
    torch.manual_seed(5)
    assert not torch.allclose(b(x), yb)


def test_hip_batchnorm_patch_leaves_cpu_and_eval_forwards_alone():
    """hip_batchnorm(model): the training-mode fusion (BatchNorm + residual + ReLU in libbtx launches) applies to CUDA tensors of a
    model in train(); on the CPU and in eval() the patched blocks run the forward they had; state_dict keys do not change; a
    user's block that merely has the same attribute names is not touched"""
    from bayesian_torch_amd.models.resnet import resnet18, BasicBlock
    from bayesian_torch_amd.models.fuse import hip_batchnorm
    torch.manual_seed(0)
    a = resnet18(num_classes=10)
    torch.manual_seed(0)
    b = resnet18(num_classes=10)
    keys = list(b.state_dict().keys())
    assert hip_batchnorm(b) == 20 and hip_batchnorm(b) == 0   # idempotent
    assert list(b.state_dict().keys()) == keys
    x = torch.randn(2, 3, 224, 224)
    a.train(); b.train()
    assert torch.equal(a(x), b(x)) and torch.equal(a.bn1.running_var, b.bn1.running_var)
    a.eval(); b.eval()
    assert torch.equal(a(x), b(x))

    class Mine(BasicBlock):   # same attributes, another forward (pre-activation): must keep its own forward
        def forward(self, x):
            return self.conv2(self.relu(self.bn2(self.conv1(self.relu(self.bn1(x)))))) + x
    m = torch.nn.Sequential(Mine(8, 8))
    assert hip_batchnorm(m) == 2 and "_btx_fwd_eval" not in m[0].__dict__


def test_fused_forms_are_chosen_by_running_the_block_not_by_its_class_name():
    """fuse_resnet / hip_batchnorm decide by a behavioural probe (models/fuse.py block_is_textbook): a user's own block class with the
    ResNet dataflow is fused, a pre-activation block with the same attribute names is left alone (and warned about) — its outputs
    must not change"""
    import warnings
    import bayesian_torch_amd as bt
    from bayesian_torch_amd.models.fuse import fuse_resnet, hip_batchnorm, block_is_textbook
    params = dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type="Reparameterization",
                  moped_enable=False, moped_delta=0.5)

    class MyBlock(torch.nn.Module):               # not models.resnet / torchvision: same dataflow, identity evaluated last
        def __init__(self, c, pre):
            super().__init__()
            self.conv1, self.bn1 = torch.nn.Conv2d(c, c, 3, padding=1, bias=False), torch.nn.BatchNorm2d(c)
            self.conv2, self.bn2 = torch.nn.Conv2d(c, c, 3, padding=1, bias=False), torch.nn.BatchNorm2d(c)
            self.relu, self.downsample, self.pre = torch.nn.ReLU(), None, pre

        def forward(self, x):
            if self.pre:                          # pre-activation: bn -> relu -> conv
                return self.conv2(self.relu(self.bn2(self.conv1(self.relu(self.bn1(x)))))) + x
            y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
            return self.relu(y + x)

    torch.manual_seed(0)
    net = torch.nn.Sequential(MyBlock(8, False), MyBlock(8, True))
    bt.dnn_to_bnn(net, params)
    net.eval()
    for bn in (net[0].bn1, net[0].bn2, net[1].bn1, net[1].bn2):
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(2, 8, 6, 6)
    torch.manual_seed(3)
    ref = net(x)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert block_is_textbook(net[0]) and not block_is_textbook(net[1])
        assert fuse_resnet(net) == 1              # only the textbook block
        assert any("left unfused" in str(i.message) for i in w)
    assert "_f1" in net[0].__dict__ and "_f1" not in net[1].__dict__
    torch.manual_seed(3)
    assert torch.allclose(net(x), ref, rtol=1e-5, atol=1e-6)   # CPU: the folded form == BN after conv; the other block untouched
    assert hip_batchnorm(net) == 4
    assert "_btx_fwd_eval" in net[0].__dict__ and "_btx_fwd_eval" not in net[1].__dict__


def test_hip_batchnorm_survives_deepcopy_pickle_and_skips_sync_batchnorm():
    """a deep copy of a patched model (EMA / AveragedModel, eval copies) must normalise with ITS OWN weights and running
    estimates, torch.save(model) must work, nn.SyncBatchNorm (cross-rank statistics) and the Lazy* variants keep torch's forward"""
    import copy
    import io
    from bayesian_torch_amd.models.fuse import hip_batchnorm
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.MaxPool2d(2))
    assert hip_batchnorm(net) == 1
    cp = copy.deepcopy(net)
    assert cp[1].forward.__self__ is cp[1] and cp[2].forward.__self__ is cp[2]
    with torch.no_grad():
        cp[1].weight.mul_(3.0)
    x = torch.randn(4, 3, 8, 8)
    net.train(); cp.train()
    rv = net[1].running_var.clone()
    y_cp = cp(x)
    assert torch.equal(net[1].running_var, rv)             # the copy's training forward leaves the original's estimates alone
    assert not torch.equal(cp[1].running_var, rv)
    assert not torch.allclose(y_cp, net(x))                 # and uses its own weight
    buf = io.BytesIO()
    torch.save(net, buf)                                    # no local closures in the patched forwards
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    net.eval(); back.eval()
    assert torch.equal(back(x), net(x))
    sync = torch.nn.Sequential(torch.nn.SyncBatchNorm(8), torch.nn.LazyBatchNorm2d())
    assert hip_batchnorm(sync) == 0 and "forward" not in sync[0].__dict__
    from bayesian_torch_amd.autograd import bn_train_usable
    assert not bn_train_usable(torch.nn.SyncBatchNorm(8), torch.randn(2, 8, 4, 4))


def test_deepcopy_gets_its_own_noise_identity_and_prior_detection():
    import copy
    from bayesian_torch_amd import layers as L
    a = L.Conv2dFlipout(8, 8, 3)
    b = copy.deepcopy(a)
    assert b._btx_layer_id != a._btx_layer_id
    assert torch.equal(a.mu_kernel, b.mu_kernel) and b.mu_kernel.data_ptr() != a.mu_kernel.data_ptr()
    m = copy.deepcopy(torch.nn.Sequential(a, torch.nn.ReLU()))
    assert m[0]._btx_layer_id not in (a._btx_layer_id, b._btx_layer_id)
    assert a._priors_are_scalar()
    a.prior_weight_mu.copy_(torch.randn_like(a.prior_weight_mu))      # in-place MOPED-style write
    assert not a._priors_are_scalar()
    a.prior_weight_mu.fill_(a.prior_mean)
    assert a._priors_are_scalar()
    a.prior_weight_mu = torch.ones_like(a.prior_weight_mu)             # buffer re-assignment (reference MOPED)
    assert not a._priors_are_scalar()


def test_presample_skips_layers_the_dma_kernels_do_not_take():
    """advisor: a depthwise / odd grouped conv queued for btx_sample_weights fails the whole batch (K % 4 != 0)"""
    from bayesian_torch_amd import layers as L
    dw = L.Conv2dFlipout(16, 16, 3, padding=1, groups=16)
    dw._btx_last_xshape = (2, 16, 8, 8)
    assert dw.presample_item(0, "bf16") is None and dw.presample_item(0, "f32") is None
    g2 = L.Conv2dFlipout(48, 48, 3, padding=1, groups=2)                # C/groups = 24: register-staged kernel
    g2._btx_last_xshape = (2, 48, 8, 8)
    assert g2.presample_item(0, "bf16") is None
    ok = L.Conv2dFlipout(64, 64, 3, padding=1)
    ok._btx_last_xshape = (2, 64, 8, 8)
    assert ok.presample_item(0, "bf16") is not None and ok.presample_item(0, "f32") is not None


LSTM_CASES = [("lstm_reparam", "LSTMReparameterization", dict(in_features=12, out_features=10), 11, 22),
              ("lstm_flipout", "LSTMFlipout", dict(in_features=16, out_features=8, bias=False), 33, 44)]


@pytest.mark.parametrize("name,cls,kw,s_init,s_fwd", LSTM_CASES)
def test_lstm_wrappers_match_the_reference(name, cls, kw, s_init, s_fwd):
    """LSTM{Reparameterization,Flipout} (reference rnn_variational.py:46-153, rnn_flipout.py:46-153) on CPU: same init
    draws, same per-step noise draws, same outputs as the reference for the same torch seeds (tests/golden/lstm.npz,
    tools/make_golden_lstm.py) — and dnn_to_bnn converts nn.LSTM instead of raising"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lstm.npz"))
    torch.manual_seed(s_init)
    layer = getattr(L, cls)(**kw)
    sd = layer.state_dict()
    ref_sd = {k[len(name) + 4:]: z[k] for k in z.files if k.startswith(name + "/sd/")}
    assert set(sd) == set(ref_sd)
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), ref_sd[k]), k
    x = torch.from_numpy(z[name + "/x"])
    torch.manual_seed(s_fwd)
    with torch.no_grad():
        hs, (hs2, cs), kl = layer(x)
    assert np.array_equal(hs.numpy(), z[name + "/hidden"]) and np.array_equal(cs.numpy(), z[name + "/cells"])
    assert float(kl) == float(z[name + "/kl"]) and float(layer.kl_loss()) == float(z[name + "/kl_loss"])
    m = torch.nn.Sequential(torch.nn.LSTM(kw["in_features"], kw["out_features"]))
    bt.dnn_to_bnn(m, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0,
                          type=cls[4:], moped_enable=False, moped_delta=0.5))
    assert m[0].__class__.__name__ == cls and m[0].dnn_to_bnn_flag
    assert len(m[0](x)) == 2  # converted layers return no KL
