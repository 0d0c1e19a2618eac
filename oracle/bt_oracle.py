"""ctypes wrapper over oracle/c/bt_oracle.c (plain-C restatement of the reference arithmetic + BTX-RNG v1).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  numpy in / numpy out, everything f32, layouts are the C-ABI's:
activations channels-last ``[NB, D, H, W, C]``, weights ``[N, taps, Cg]``, outputs ``[NB, Do, Ho, Wo, N]``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Geom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "NB", "D", "H", "W", "C", "N", "KD", "KH", "KW", "sd", "sh", "sw", "pd", "ph", "pw",
        "dd", "dh", "dw", "od", "oh", "ow", "groups")]


def build(force=False):
    """gcc -O2 the oracle into oracle/libbt_oracle.so (seconds)."""
    so = os.path.join(_HERE, "libbt_oracle.so")
    src = os.path.join(_HERE, "c", "bt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        i8p = ctypes.POINTER(ctypes.c_int8)
        L.bto_softplus.restype = ctypes.c_float
        L.bto_softplus.argtypes = [ctypes.c_float]
        L.bto_bf16_round.restype = ctypes.c_float
        L.bto_bf16_round.argtypes = [ctypes.c_float]
        L.bto_kl_mean.restype = ctypes.c_double
        L.bto_kl_mean.argtypes = [fp, fp, ctypes.c_size_t, fp, fp, ctypes.c_float, ctypes.c_float]
        L.bto_eps.restype = None
        L.bto_eps.argtypes = [fp, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.bto_sign.restype = None
        L.bto_sign.argtypes = [i8p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.bto_philox.restype = None
        L.bto_philox.argtypes = [ctypes.c_uint32] * 6 + [ctypes.POINTER(ctypes.c_uint32)]
        L.bto_out_shape.restype = ctypes.c_int
        L.bto_out_shape.argtypes = [ctypes.POINTER(Geom), ctypes.c_int] + [ctypes.POINTER(ctypes.c_int32)] * 3
        L.bto_contract_fwd.restype = ctypes.c_int
        L.bto_contract_fwd.argtypes = [ctypes.c_int, ctypes.POINTER(Geom), ctypes.c_int, fp, fp, fp, fp, fp, fp, fp,
                                       i8p, i8p, ctypes.c_int, fp]
        L.bto_mc_accumulate.restype = None
        L.bto_mc_accumulate.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                        ctypes.POINTER(ctypes.c_double)]
        _LIB = L
    return _LIB


def _fp(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i8(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_int8))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def softplus(rho):
    return np.log1p(np.exp(np.asarray(rho, dtype=np.float32))).astype(np.float32)


def kl_mean(mu, rho, prior_mu=0.0, prior_sigma=1.0, prior_mu_t=None, prior_sigma_t=None):
    mu, rho = _f32(mu).ravel(), _f32(rho).ravel()
    pm, ps = _f32(prior_mu_t), _f32(prior_sigma_t)
    if pm is not None:
        pm = pm.ravel()
    if ps is not None:
        ps = ps.ravel()
    return float(lib().bto_kl_mean(_fp(mu), _fp(rho), mu.size, _fp(pm), _fp(ps), prior_mu, prior_sigma))


def philox(c, k):
    out = (ctypes.c_uint32 * 4)()
    lib().bto_philox(*[int(v) & 0xFFFFFFFF for v in c], *[int(v) & 0xFFFFFFFF for v in k], out)
    return [int(v) for v in out]


def eps(n, seed, sample, layer, stream):
    out = np.empty(int(n), dtype=np.float32)
    lib().bto_eps(_fp(out), out.size, int(seed), int(sample), int(layer), int(stream))
    return out


def sign(n, seed, sample, layer, stream):
    out = np.empty(int(n), dtype=np.int8)
    lib().bto_sign(_i8(out), out.size, int(seed), int(sample), int(layer), int(stream))
    return out


def bf16_round(a):
    """round-to-nearest-even to bf16, returned as f32 (numpy restatement of bto_bf16_round)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).copy()
    nan = np.isnan(a)
    out[nan] = a[nan]
    return out


def make_geom(NB, spatial, C, N, kernel, stride, padding, dilation, groups=1, output_padding=(0, 0, 0)):
    """spatial/kernel/stride/... are 3-tuples (D,H,W order); lower-rank convs pad with leading 1s / 0s."""
    g = Geom()
    g.NB, (g.D, g.H, g.W), g.C, g.N = NB, spatial, C, N
    g.KD, g.KH, g.KW = kernel
    g.sd, g.sh, g.sw = stride
    g.pd, g.ph, g.pw = padding
    g.dd, g.dh, g.dw = dilation
    g.od, g.oh, g.ow = output_padding
    g.groups = groups
    return g


def out_shape(g, transposed=False):
    d, h, w = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    rc = lib().bto_out_shape(ctypes.byref(g), int(transposed), ctypes.byref(d), ctypes.byref(h), ctypes.byref(w))
    if rc:
        raise ValueError("bad geometry")
    return d.value, h.value, w.value


def contract_fwd(kind, g, x, mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, sign_in=None, sign_out=None,
                 transposed=False, bf16_inputs=False):
    """x [NB,D,H,W,C]; mu_w/rho_w/eps_w [N,taps,Cg]; returns out [NB,Do,Ho,Wo,N] f32."""
    Do, Ho, Wo = out_shape(g, transposed)
    x, mu_w, rho_w, eps_w = _f32(x), _f32(mu_w), _f32(rho_w), _f32(eps_w)
    mu_b, rho_b, eps_b = _f32(mu_b), _f32(rho_b), _f32(eps_b)
    si = None if sign_in is None else np.ascontiguousarray(sign_in, dtype=np.int8)
    so = None if sign_out is None else np.ascontiguousarray(sign_out, dtype=np.int8)
    out = np.empty((g.NB, Do, Ho, Wo, g.N), dtype=np.float32)
    rc = lib().bto_contract_fwd(int(kind), ctypes.byref(g), int(transposed), _fp(x), _fp(mu_w), _fp(rho_w), _fp(mu_b),
                                _fp(rho_b), _fp(eps_w), _fp(eps_b), _i8(si), _i8(so), int(bf16_inputs), _fp(out))
    if rc:
        raise RuntimeError("bto_contract_fwd rc=%d" % rc)
    return out


def mc_accumulate(logits, kl, packed=None):
    logits = _f32(logits)
    bs, C = logits.shape
    if packed is None:
        packed = np.zeros(2 * bs * C + bs + 2, dtype=np.float64)
    lib().bto_mc_accumulate(_fp(logits), bs, C, float(kl), packed.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return packed
