"""The oracle pinned against vectors generated from the reference itself (tests/golden, tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from helpers import case_geometry, oracle_forward, rel_l2
from oracle import bt_oracle as o
from oracle import bt_ref


def test_philox_known_answer():
    # Random123 kat_vectors: philox4x32 10 rounds
    assert o.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert o.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert o.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_c_oracle_matches_reference_outputs(golden):
    """plain-C restatement (f64 accumulation) vs the reference's f32 outputs with the reference's own noise"""
    for name, (meta, d) in golden["cases"].items():
        geo = case_geometry(meta)
        out = oracle_forward(geo, d["x"], d["mu_w"], d["rho_w"], d.get("mu_b"), d.get("rho_b"), d["eps_w"],
                             d.get("eps_b"), d.get("sign_in"), d.get("sign_out"))
        assert out.shape == d["out"].shape, name
        assert rel_l2(out, d["out"]) < 2e-6, (name, rel_l2(out, d["out"]))


def test_c_oracle_kl_matches_reference(golden):
    for name, (meta, d) in golden["cases"].items():
        kl = o.kl_mean(d["mu_w"], d["rho_w"])
        if "mu_b" in d:
            kl += o.kl_mean(d["mu_b"], d["rho_b"])
        assert abs(kl - meta["kl"]) <= 2e-6 * abs(meta["kl"]), name


def test_torch_restatement_bit_exact(golden):
    """oracle/bt_ref.py (the cpu_baseline port) reproduces the reference outputs BIT-exactly from the stored noise"""
    for name, (meta, d) in golden["cases"].items():
        geo = case_geometry(meta)
        nd = geo["nd"]
        if nd == 0:
            op = dict(kind="linear")
        else:
            op = dict(kind="convT" if geo["transposed"] else "conv", nd=nd, stride=geo["stride"][3 - nd:],
                      padding=geo["padding"][3 - nd:], dilation=geo["dilation"][3 - nd:], groups=geo["groups"],
                      output_padding=geo["outpad"][3 - nd:])
        t = {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}
        args = (t["x"], t["mu_w"], t["rho_w"], t.get("mu_b"), t.get("rho_b"), t["eps_w"], t.get("eps_b"))
        with torch.no_grad():
            if geo["kind"] == 0:
                out = bt_ref.reparam_forward(*args, op)
            else:
                out = bt_ref.flipout_forward(*args, t["sign_in"].float(), t["sign_out"].float(), op)
            kl = bt_ref.kl_loss(t["mu_w"], t["rho_w"], t.get("mu_b"), t.get("rho_b"))
        assert torch.equal(out, t["out"]), name
        assert float(kl) == meta["kl"], name


def test_rng_statistics():
    """BTX-RNG v1 is not the reference's generator; it must still be N(0,1) / fair signs."""
    e = o.eps(1 << 18, 1234, 3, 7, 0).astype(np.float64)
    assert abs(e.mean()) < 0.01 and abs(e.std() - 1.0) < 0.01
    assert abs((e ** 3).mean()) < 0.03 and abs((e ** 4).mean() - 3.0) < 0.08
    assert np.abs(e).max() < 6.0
    s = o.sign(1 << 18, 1234, 3, 7, 2).astype(np.float64)
    assert set(np.unique(s)) == {-1.0, 1.0}
    assert abs(s.mean()) < 0.01
    # neighbouring elements, neighbouring samples and the two sign streams are uncorrelated
    assert abs((s[1:] * s[:-1]).mean()) < 0.01
    s2 = o.sign(1 << 18, 1234, 4, 7, 2).astype(np.float64)
    s3 = o.sign(1 << 18, 1234, 3, 7, 3).astype(np.float64)
    assert abs((s * s2).mean()) < 0.01 and abs((s * s3).mean()) < 0.01
    e2 = o.eps(1 << 18, 1234, 4, 7, 0).astype(np.float64)
    assert abs((e * e2).mean()) < 0.01
    # per-bit-position balance of the 32-sign words
    w = s.reshape(-1, 32)
    assert np.abs(w.mean(axis=0)).max() < 0.05


def test_bf16_round():
    x = np.array([1.0, 1.00390625, 1.005859375, -3.14159, 1e-40, 65504.0], dtype=np.float32)
    r = o.bf16_round(x)
    t = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(r, t)
    assert all(o.lib().bto_bf16_round(float(v)) == float(w) for v, w in zip(x, t))


def test_mc_accumulate_oracle():
    rng = np.random.default_rng(0)
    lg = rng.normal(size=(4, 10)).astype(np.float32)
    p = o.mc_accumulate(lg, 2.5)
    sm = torch.softmax(torch.from_numpy(lg).double(), 1).numpy()
    assert np.allclose(p[:40], sm.ravel()) and np.allclose(p[40:80], (sm ** 2).ravel())
    assert np.allclose(p[80:84], -(sm * np.log(sm + 1e-15)).sum(1))
    assert p[84] == 2.5 and p[85] == 1.0
