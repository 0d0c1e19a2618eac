"""Shared test helpers: golden loading and logical <-> C-ABI layout conversion (numpy, independent of the product's
own packing code so the oracle side of every comparison stays independent)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_golden():
    z = np.load(os.path.join(HERE, "golden", "layers.npz"))
    with open(os.path.join(HERE, "golden", "kat.json")) as f:
        kat = json.load(f)
    cases = {}
    for name, meta in kat["layers"].items():
        d = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}
        cases[name] = (meta, d)
    return dict(cases=cases, kat=kat)


def case_geometry(meta):
    """-> dict(nd, transposed, kind, Cin, Cout, kernel3, stride3, padding3, dilation3, outpad3, groups)"""
    cls, kw = meta["cls"], meta["kwargs"]
    kind = 1 if "Flipout" in cls else 0
    if cls.startswith("Linear"):
        return dict(nd=0, transposed=False, kind=kind, Cin=kw["in_features"], Cout=kw["out_features"],
                    kernel=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), outpad=(0, 0, 0),
                    groups=1)
    nd = int(cls[cls.index("d") - 1])

    def t3(v, fill):
        v = tuple(v) if isinstance(v, (list, tuple)) else (v,) * nd
        return (fill,) * (3 - nd) + v
    return dict(nd=nd, transposed="Transpose" in cls, kind=kind, Cin=kw["in_channels"], Cout=kw["out_channels"],
                kernel=t3(kw["kernel_size"], 1), stride=t3(kw.get("stride", 1), 1), padding=t3(kw.get("padding", 0), 0),
                dilation=t3(kw.get("dilation", 1), 1), outpad=t3(kw.get("output_padding", 0), 0),
                groups=kw.get("groups", 1))


def to_cl(x, nd):
    """logical [N,C,*sp] (or [*,C] for nd=0) -> channels-last [NB,D,H,W,C] numpy"""
    x = np.asarray(x)
    if nd == 0:
        x2 = x.reshape(-1, x.shape[-1])
        return np.ascontiguousarray(x2.reshape(x2.shape[0], 1, 1, 1, x2.shape[1]))
    perm = (0,) + tuple(range(2, 2 + nd)) + (1,)
    xc = np.transpose(x, perm)
    sp = xc.shape[1:-1]
    return np.ascontiguousarray(xc.reshape((xc.shape[0],) + (1,) * (3 - nd) + sp + (xc.shape[-1],)))


def from_cl(o, nd, lead_shape=None):
    """channels-last [NB,Do,Ho,Wo,N] -> logical [N,C,*sp] (nd>0) or [*lead, C]"""
    if nd == 0:
        o2 = o.reshape(o.shape[0], o.shape[-1])
        return o2 if lead_shape is None else o2.reshape(tuple(lead_shape) + (o.shape[-1],))
    sp = o.shape[4 - nd:4]
    o2 = o.reshape((o.shape[0],) + sp + (o.shape[-1],))
    perm = (0, nd + 1) + tuple(range(1, nd + 1))
    return np.ascontiguousarray(np.transpose(o2, perm))


def w_to_gemm(w, geo):
    """logical parameter tensor -> [N, taps, Cg]"""
    w = np.asarray(w)
    nd = geo["nd"]
    if nd == 0:
        return np.ascontiguousarray(w.reshape(w.shape[0], 1, w.shape[1]))
    if not geo["transposed"]:
        perm = (0,) + tuple(range(2, 2 + nd)) + (1,)
        t = np.transpose(w, perm)
        return np.ascontiguousarray(t.reshape(w.shape[0], -1, w.shape[1]))
    g = geo["groups"]
    cin, ng = w.shape[0], w.shape[1]
    t = w.reshape(g, cin // g, ng, -1).transpose(0, 2, 3, 1)  # [g, Ng, T, Cg]
    return np.ascontiguousarray(t.reshape(g * ng, -1, cin // g))


def oracle_geom(geo, x_logical_shape):
    from oracle import bt_oracle as o
    nd = geo["nd"]
    if nd == 0:
        nb = int(np.prod(x_logical_shape[:-1]))
        spatial = (1, 1, 1)
    else:
        nb = x_logical_shape[0]
        spatial = (1,) * (3 - nd) + tuple(x_logical_shape[2:])
    return o.make_geom(nb, spatial, geo["Cin"], geo["Cout"], geo["kernel"], geo["stride"], geo["padding"],
                       geo["dilation"], geo["groups"], geo["outpad"])


def oracle_forward(geo, x, mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, sign_in=None, sign_out=None, bf16=False):
    """All arguments in the reference's LOGICAL layouts (numpy); returns logical output (numpy f32)."""
    from oracle import bt_oracle as o
    nd = geo["nd"]
    g = oracle_geom(geo, np.shape(x))
    si = None if sign_in is None else to_cl(sign_in, nd).astype(np.int8)
    so = None if sign_out is None else to_cl(sign_out, nd).astype(np.int8)
    out = o.contract_fwd(geo["kind"], g, to_cl(x, nd), w_to_gemm(mu_w, geo), w_to_gemm(rho_w, geo), mu_b, rho_b,
                         w_to_gemm(eps_w, geo), eps_b, si, so, transposed=geo["transposed"], bf16_inputs=bf16)
    return from_cl(out, nd, None if nd else np.shape(x)[:-1])


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
