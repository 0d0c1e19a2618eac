#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE ITSELF (IntelLabs/bayesian-torch at /root/reference).

The reference has no tests or golden vectors of its own, so parity is pinned by running the
reference here (it imports on torch-CPU) and freezing small input/output vectors.  This script only runs in the
build container — /root/reference does not exist on the GPU box; the fixtures it writes are committed.

For every case:  torch.manual_seed(init_seed) -> reference layer (its own init draws) ; x ~ randn ;
torch.manual_seed(fwd_seed) -> out, kl = layer(x)  (the reference draws eps / signs from the global generator).
The noise the reference consumed is re-drawn with the same seed in the reference's order
(oracle/bt_ref.draw_noise_like_reference) and the torch restatement is asserted BIT-EXACT against the reference
output before anything is written — so the stored (eps, signs) are exactly what produced `out`.

usage: python tools/make_golden.py      (writes tests/golden/layers.npz, tests/golden/kat.json)
"""
import inspect
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, ROOT)

import bayesian_torch.layers as RL  # noqa: E402  (the reference)
from bayesian_torch.models.dnn_to_bnn import dnn_to_bnn as ref_dnn_to_bnn, get_kl_loss as ref_get_kl_loss  # noqa: E402
from bayesian_torch.models.deterministic import resnet_large as ref_resnet  # noqa: E402
from bayesian_torch.utils.util import get_rho as ref_get_rho  # noqa: E402
from oracle import bt_ref  # noqa: E402

CASES = [
    # name, class, ctor kwargs, input shape
    ("linear_reparam_cfg0", "LinearReparameterization", dict(in_features=128, out_features=64), (32, 128)),
    ("linear_reparam_nobias", "LinearReparameterization", dict(in_features=40, out_features=24, bias=False), (5, 40)),
    ("linear_flipout", "LinearFlipout", dict(in_features=96, out_features=80), (16, 96)),
    ("linear_flipout_odd", "LinearFlipout", dict(in_features=50, out_features=10), (7, 50)),
    ("conv2d_reparam", "Conv2dReparameterization",
     dict(in_channels=16, out_channels=32, kernel_size=3, stride=1, padding=1), (2, 16, 12, 12)),
    ("conv2d_reparam_stem", "Conv2dReparameterization",
     dict(in_channels=3, out_channels=16, kernel_size=7, stride=2, padding=3, bias=False), (2, 3, 32, 32)),
    ("conv2d_flipout", "Conv2dFlipout",
     dict(in_channels=16, out_channels=32, kernel_size=3, stride=2, padding=1), (2, 16, 14, 14)),
    ("conv2d_flipout_nobias_c64", "Conv2dFlipout",
     dict(in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=False), (1, 64, 8, 8)),
    ("conv2d_flipout_groups_dil", "Conv2dFlipout",
     dict(in_channels=16, out_channels=24, kernel_size=3, stride=1, padding=2, dilation=2, groups=2), (2, 16, 9, 11)),
    ("conv2d_flipout_1x1_s2", "Conv2dFlipout",
     dict(in_channels=32, out_channels=48, kernel_size=1, stride=2, padding=0, bias=False), (2, 32, 10, 10)),
    ("conv1d_reparam", "Conv1dReparameterization",
     dict(in_channels=8, out_channels=12, kernel_size=5, stride=2, padding=2), (3, 8, 33)),
    ("conv1d_flipout", "Conv1dFlipout",
     dict(in_channels=8, out_channels=16, kernel_size=3, stride=1, padding=1), (2, 8, 20)),
    ("conv3d_reparam", "Conv3dReparameterization",
     dict(in_channels=4, out_channels=8, kernel_size=3, prior_mean=0, prior_variance=1, posterior_mu_init=0,
          posterior_rho_init=-3.0, stride=1, padding=1), (2, 4, 5, 6, 7)),
    ("conv3d_flipout", "Conv3dFlipout",
     dict(in_channels=8, out_channels=8, kernel_size=(1, 3, 3), stride=(1, 2, 1), padding=(0, 1, 1)), (1, 8, 3, 8, 6)),
    ("convT2d_reparam", "ConvTranspose2dReparameterization",
     dict(in_channels=8, out_channels=12, kernel_size=3, stride=2, padding=1, output_padding=1), (2, 8, 6, 7)),
    ("convT2d_flipout", "ConvTranspose2dFlipout",
     dict(in_channels=8, out_channels=8, kernel_size=4, stride=2, padding=1), (2, 8, 5, 5)),
    ("convT1d_reparam", "ConvTranspose1dReparameterization",
     dict(in_channels=6, out_channels=10, kernel_size=3, stride=3, padding=0), (2, 6, 9)),
    ("convT3d_reparam", "ConvTranspose3dReparameterization",
     dict(in_channels=4, out_channels=4, kernel_size=2, stride=2, padding=0), (1, 4, 3, 3, 3)),
    # round 2: shapes the FAST kernels take (C/groups % 32 == 0), so the reference's own noise reaches the tap-unrolled
    # patch kernel, the run-time-tap patch kernel and the LDS-DMA kernel (explicit eps_w / sign_in / sign_out)
    ("conv2d_reparam_c32_3x3", "Conv2dReparameterization",
     dict(in_channels=32, out_channels=64, kernel_size=3, stride=1, padding=1), (2, 32, 10, 10)),
    ("conv2d_flipout_c32_3x3_s2", "Conv2dFlipout",
     dict(in_channels=32, out_channels=32, kernel_size=3, stride=2, padding=1), (2, 32, 11, 11)),
    ("conv2d_flipout_c32_5x5", "Conv2dFlipout",
     dict(in_channels=32, out_channels=32, kernel_size=5, stride=1, padding=2, bias=False), (1, 32, 8, 8)),
    ("conv2d_flipout_c64_3x3_multi", "Conv2dFlipout",
     dict(in_channels=64, out_channels=96, kernel_size=3, stride=1, padding=1), (3, 64, 13, 9)),
]


def op_of(layer, cls):
    if cls.startswith("Linear"):
        return dict(kind="linear")
    nd = int(cls[cls.index("d") - 1])
    tup = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v,) * nd  # noqa: E731
    op = dict(kind="convT" if "Transpose" in cls else "conv", nd=nd, stride=tup(layer.stride),
              padding=tup(layer.padding), dilation=tup(layer.dilation), groups=layer.groups)
    if "Transpose" in cls:
        op["output_padding"] = tup(layer.output_padding)
    return op


def main():
    out = {}
    meta = {}
    for i, (name, cls, kw, xshape) in enumerate(CASES):
        init_seed, fwd_seed = 100 + i, 9000 + i
        torch.manual_seed(init_seed)
        layer = getattr(RL, cls)(**kw)
        x = torch.randn(*xshape)
        wn = "weight" if cls.startswith("Linear") else "kernel"
        mu_w, rho_w = getattr(layer, "mu_" + wn).data, getattr(layer, "rho_" + wn).data
        has_b = layer.mu_bias is not None
        mu_b = layer.mu_bias.data if has_b else None
        rho_b = layer.rho_bias.data if has_b else None
        with torch.no_grad():
            torch.manual_seed(fwd_seed)
            ref_out, ref_kl = layer(x)
            ref_kl2 = layer.kl_loss()
        assert float(ref_kl) == float(ref_kl2)
        fam = "reparam" if "Reparameterization" in cls else ("linear_flipout" if cls.startswith("Linear")
                                                            else "conv_flipout")
        op = op_of(layer, cls)
        torch.manual_seed(fwd_seed)
        eps_w, eps_b, s_in, s_out = bt_ref.draw_noise_like_reference(
            fam, tuple(x.shape), tuple(ref_out.shape), tuple(mu_w.shape), mu_b.shape[0] if has_b else 0)
        with torch.no_grad():
            if fam == "reparam":
                re = bt_ref.reparam_forward(x, mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, op)
            else:
                re = bt_ref.flipout_forward(x, mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, s_in, s_out, op)
            rkl = bt_ref.kl_loss(mu_w, rho_w, mu_b, rho_b, layer.prior_mean, layer.prior_variance)
        assert torch.equal(re, ref_out), "%s: restatement is not bit-exact vs the reference (max %g)" % (
            name, (re - ref_out).abs().max())
        assert float(rkl) == float(ref_kl), name
        out[name + "/x"] = x.numpy()
        out[name + "/mu_w"] = mu_w.numpy()
        out[name + "/rho_w"] = rho_w.numpy()
        if has_b:
            out[name + "/mu_b"] = mu_b.numpy()
            out[name + "/rho_b"] = rho_b.numpy()
            out[name + "/eps_b"] = eps_b.numpy()
        out[name + "/eps_w"] = eps_w.numpy()
        if s_in is not None:
            assert (s_in != 0).all() and (s_out != 0).all()
            out[name + "/sign_in"] = s_in.numpy().astype(np.int8)
            out[name + "/sign_out"] = s_out.numpy().astype(np.int8)
        out[name + "/out"] = ref_out.numpy()
        meta[name] = dict(cls=cls, kwargs={k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                          x_shape=list(xshape), init_seed=init_seed, fwd_seed=fwd_seed, kl=float(ref_kl),
                          signature=str(inspect.signature(getattr(RL, cls).__init__)),
                          state_dict_keys=list(layer.state_dict().keys()))
        print("%-28s out %-18s kl %.9g  bit-exact restatement OK" % (name, tuple(ref_out.shape), float(ref_kl)))

    # signatures of every class on the path (API surface pin)
    sigs = {}
    for n in sorted(dir(RL)):
        c = getattr(RL, n)
        if inspect.isclass(c) and ("Reparameterization" in n or "Flipout" in n) and "Quant" not in n and "LSTM" not in n:
            sigs[n] = str(inspect.signature(c.__init__))

    # known-answer KL values for whole models
    kat = {}
    base = dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0,
                moped_enable=False, moped_delta=0.5)
    for arch, typ, moped in (("resnet18", "Reparameterization", False), ("resnet18", "Flipout", False),
                             ("resnet18", "Flipout", True), ("resnet50", "Flipout", False), ("resnet50", "Flipout", True)):
        torch.manual_seed(0)
        m = getattr(ref_resnet, arch)()
        p = dict(base, type=typ, moped_enable=moped)
        ref_dnn_to_bnn(m, p)
        kl = float(ref_get_kl_loss(m))
        names = [(n, mod.__class__.__name__) for n, mod in m.named_modules() if hasattr(mod, "kl_loss")]
        kat["%s_%s%s" % (arch, typ, "_moped" if moped else "")] = dict(
            kl=kl, n_bayes_layers=len(names), first=names[0], last=names[-1],
            checksum_mu=float(sum(float(q.double().sum()) for n, q in m.named_parameters() if ".mu_" in n or n.startswith("mu_"))))
        print(arch, typ, "moped" if moped else "", "kl =", repr(kl))
    w = torch.linspace(-0.3, 0.3, 13)
    kat["get_rho"] = dict(w=w.tolist(), delta=0.5, rho=ref_get_rho(w, 0.5).tolist())

    # utils.util.MOPED(): tensor-valued priors + posterior init from a deterministic model (reference utils/util.py:72-136)
    import tempfile
    from bayesian_torch.utils.util import MOPED as ref_MOPED

    def small_net():
        return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                                   torch.nn.Conv2d(8, 16, 3, stride=2, bias=False), torch.nn.Flatten(),
                                   torch.nn.Linear(16 * 7 * 7, 10))
    torch.manual_seed(0)
    det = small_net()
    det[1].running_mean.normal_()
    det[1].running_var.uniform_(0.5, 2.0)
    torch.manual_seed(1)
    bnn = small_net()
    for typ in ("Reparameterization", "Flipout"):
        torch.manual_seed(1)
        bnn = small_net()
        ref_dnn_to_bnn(bnn, dict(base, type=typ))
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            torch.save(det.state_dict(), f.name)
            ref_MOPED(bnn, small_net(), f.name, 0.1)
        kat["moped_fn_" + typ] = dict(kl=float(ref_get_kl_loss(bnn)), delta=0.1,
                                      bn_mean_sum=float(bnn[1].running_mean.sum()))
        print("MOPED()", typ, "kl =", repr(kat["moped_fn_" + typ]["kl"]))

    gdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gdir, exist_ok=True)

    # distribution of the reference's MC samples: BASELINE cfg2 MLP, 256 stochastic forwards of the
    # reference on torch-CPU; per-logit mean / variance of the first 32 batch rows
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(784, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(),
                              torch.nn.Linear(512, 10))
    ref_dnn_to_bnn(net, dict(base, type="Flipout"))
    net.eval()
    torch.manual_seed(1234)
    xm = torch.randn(256, 784)
    S, rows = 256, 32
    torch.manual_seed(4321)
    with torch.no_grad():
        ys = torch.stack([net(xm)[:rows] for _ in range(S)]).double()
    np.savez_compressed(os.path.join(gdir, "mc_stats.npz"), mean=ys.mean(0).numpy(), var=ys.var(0, unbiased=False).numpy(),
                        S=S, rows=rows)
    print("mc_stats: mean |logit| %.3g, mean std %.3g" % (float(ys.mean(0).abs().mean()), float(ys.std(0).mean())))
    np.savez_compressed(os.path.join(gdir, "layers.npz"), **out)
    with open(os.path.join(gdir, "kat.json"), "w") as f:
        json.dump(dict(layers=meta, signatures=sigs, models=kat,
                       generated_by="tools/make_golden.py from /root/reference (bayesian-torch v0.5.0), torch %s CPU"
                       % torch.__version__), f, indent=1, sort_keys=True)
    print("wrote", gdir, "%.1f KB" % (os.path.getsize(os.path.join(gdir, "layers.npz")) / 1024))


if __name__ == "__main__":
    main()
