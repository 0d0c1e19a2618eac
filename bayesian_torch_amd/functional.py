"""Host-side plumbing between torch tensors and the C-ABI (include/btx.h).

torch is used here for device memory, streams and layout views only; all arithmetic of the hot path on a GPU
tensor happens inside libbtx.so.  CPU tensors take the ATen route (`*_aten` below: the reference's own op chain —
BASELINE.json config 0, "plumbing, no GPU"); a CUDA tensor NEVER does: if the library is missing or a case is
unsupported the call raises.
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib
from . import rng as _rng

_PRECISION = os.environ.get("BTX_PRECISION", "f32")  # "f32" (parity mode) | "bf16x3" | "bf16" (throughput mode)
PRECISIONS = ("f32", "bf16", "bf16x3")


def set_precision(prec):
    """Contraction precision of the HIP path: "f32" = v_mfma_f32_32x32x2_f32 (exact f32 fma chain, parity to
    ~1e-6 rel), "bf16" = v_mfma_f32_32x32x16_bf16 with f32 accumulation (rel-L2 ~3e-3, stated in DESIGN.md),
    "bf16x3" = split-bf16: f32 activations, operands as hi + lo bf16, three bf16 MFMAs per product (rel-L2 ~1e-6 per
    layer: inside north_star's 1e-4 at a third of the bf16 matrix rate)."""
    global _PRECISION
    if prec not in PRECISIONS:
        raise ValueError("precision must be one of %s" % (PRECISIONS,))
    _PRECISION = prec


def get_precision():
    return _PRECISION


# Successive contraction launches walk their tiles in alternating directions (BTX_FLAG_REVERSE on every other call): a
# layer then starts on the activations its producer wrote last.  Values do not depend on it.  BTX_ALT_ORDER=0 disables (A/B).
_ALT_ORDER = os.environ.get("BTX_ALT_ORDER", "1") != "0"
_ORDER_TOGGLE = [0]
_OUT_LAYOUT = "channels_last"
_CONCURRENT = False  # set by mc.GraphedMC(lanes > 1) while it captures: BTX_FLAG_CONCURRENT on every contraction launch


class concurrent_plan:
    """context manager: plan the contraction launches issued inside for device throughput (BTX_FLAG_CONCURRENT) — what
    mc.GraphedMC(lanes > 1) does while it captures.  The K-split of small-map layers, hence the f32 summation order, can
    differ from the default (latency) plan by rounding."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global _CONCURRENT
        self.prev, _CONCURRENT = _CONCURRENT, self.on
        return self

    def __exit__(self, *exc):
        global _CONCURRENT
        _CONCURRENT = self.prev
        return False


def set_output_layout(layout):
    """Memory format of the HIP path's convolution outputs.  "channels_last" (default): the kernels' native layout —
    logical [N,C,*sp] shape with channels-last strides, zero extra passes, but `out.view(N, -1)` on a map with H,W > 1
    raises like for any channels_last tensor (use `.reshape` / `.flatten`).  "contiguous": one extra HBM pass per layer
    makes the outputs plain contiguous NCHW, for user code written against the reference that calls `.view`."""
    global _OUT_LAYOUT
    if layout not in ("channels_last", "contiguous"):
        raise ValueError("layout must be 'channels_last' or 'contiguous'")
    _OUT_LAYOUT = layout


def _triple(v, nd, fill):
    """int or nd-tuple -> 3-tuple in (D, H, W) order, padded in front with `fill`."""
    if isinstance(v, int):
        v = (v,) * max(nd, 1)
    v = tuple(int(a) for a in v)
    if len(v) != max(nd, 1) and nd > 0:
        raise ValueError("expected %d values, got %r" % (nd, v))
    if nd == 0:
        return (fill, fill, fill)
    return (fill,) * (3 - nd) + v


class OpDesc:
    """Geometry of one variational contraction (Linear: nd=0)."""

    __slots__ = ("nd", "transposed", "kernel", "stride", "padding", "dilation", "output_padding", "groups",
                 "in_channels", "out_channels")

    def __init__(self, nd, in_channels, out_channels, kernel=1, stride=1, padding=0, dilation=1, groups=1,
                 transposed=False, output_padding=0):
        self.nd, self.transposed, self.groups = nd, bool(transposed), int(groups)
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel = _triple(kernel, nd, 1)
        self.stride = _triple(stride, nd, 1)
        self.padding = _triple(padding, nd, 0)
        self.dilation = _triple(dilation, nd, 1)
        self.output_padding = _triple(output_padding, nd, 0)

    def out_spatial(self, spatial):
        out = []
        for i, k, s, p, d, op in zip(spatial, self.kernel, self.stride, self.padding, self.dilation,
                                     self.output_padding):
            if self.transposed:
                out.append((i - 1) * s - 2 * p + d * (k - 1) + op + 1)
            else:
                out.append((i + 2 * p - d * (k - 1) - 1) // s + 1)
        return tuple(out)


def pack_gemm_major(w, op):
    """logical parameter tensor -> the kernel's [N][tap][Cg] f32 layout (a pure layout permute).
    Linear [out,in]: unchanged.  Conv [Cout, Cin/g, *k] -> [Cout, *k, Cin/g] (== channels_last storage).
    ConvTranspose [Cin, Cout/g, *k] -> [g, Cout/g, taps, Cin/g]."""
    if op.nd == 0:
        return w.contiguous()
    if not op.transposed:
        perm = (0,) + tuple(range(2, 2 + op.nd)) + (1,)
        return w.permute(perm).contiguous()
    g = op.groups
    cin, ng = w.shape[0], w.shape[1]
    t = w.reshape(g, cin // g, ng, -1)  # [g, Cg, Ng, T]
    return t.permute(0, 2, 3, 1).contiguous()  # [g, Ng, T, Cg]


def gemm_major_param(shape, op):
    """Allocate a parameter whose STORAGE is the kernel's [N][tap][Cg] layout and return the logical view the
    reference exposes ([out,in] / [Cout,Cin/g,*k] / [Cin,Cout/g,*k]).  The HIP path then reads the parameter in
    place — no packed copy that could go stale when user code mutates `param.data` (which does not bump
    `_version`).  ConvTranspose with groups > 1 has no strided logical view and stays contiguous (re-packed on
    every forward)."""
    nd = op.nd
    if nd == 0:
        return torch.empty(shape)
    k = tuple(shape[2:])
    if not op.transposed:
        phys = torch.empty((shape[0],) + k + (shape[1],))
        return phys.permute((0, nd + 1) + tuple(range(1, nd + 1)))
    if op.groups == 1:
        phys = torch.empty((shape[1],) + k + (shape[0],))  # [Cout][*k][Cin]
        return phys.permute((nd + 1, 0) + tuple(range(1, nd + 1)))
    return torch.empty(shape)


def gemm_major_view(w, op):
    """The [N][tap][Cg]-ordered tensor of a parameter: a zero-copy view when the parameter is stored GEMM-major
    (the default, see gemm_major_param), else a packed copy made now."""
    nd = op.nd
    w = w.detach()
    if w.dtype != torch.float32:
        raise _lib.BtxError("variational parameters must be float32 (got %s)" % w.dtype)
    if nd == 0:
        return w if w.is_contiguous() else w.contiguous()
    if not op.transposed:
        v = w.permute((0,) + tuple(range(2, 2 + nd)) + (1,))
        return v if v.is_contiguous() else v.contiguous()
    if op.groups == 1:
        v = w.permute((1,) + tuple(range(2, 2 + nd)) + (0,))
        return v if v.is_contiguous() else v.contiguous()
    return pack_gemm_major(w, op)


_WS = {}
_GEOM_CACHE = {}  # (id(op), batch, extent, kind, dtypes, flags) -> (BtxGeom, workspace bytes, op)

# optional per-launch HIP-event timing of btx_contract_fwd (bench.py's roofline leg).  Events are recorded on the
# stream the kernel is launched on (torch's current stream == the stream handed to the C-ABI).
_LAUNCH_LOG = None


def enable_launch_timing(on=True):
    global _LAUNCH_LOG
    _LAUNCH_LOG = [] if on else None


def launch_log():
    """[(tag, flops, start_event, end_event)] recorded since enable_launch_timing(True)."""
    return _LAUNCH_LOG


def _workspace(device, nbytes, stream):
    key = (device, stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def _to_channels_last(x, op):
    """-> (physical channels-last contiguous tensor, NB, (D,H,W), restore(out_phys, out_spatial) -> logical)."""
    if op.nd == 0:
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        return x2, x2.shape[0], (1, 1, 1), lambda o, sp: o.reshape(*lead, op.out_channels)
    if x.dim() == op.nd + 1:  # unbatched [C,*sp], as ATen's convolutions accept
        xp, nb, sp, restore = _to_channels_last(x.unsqueeze(0), op)
        return xp, nb, sp, lambda o, s_: restore(o, s_).squeeze(0)
    if x.dim() != op.nd + 2:
        raise ValueError("expected %dD input, got %dD" % (op.nd + 2, x.dim()))
    nb = x.shape[0]
    if op.nd == 1:
        x4 = x.unsqueeze(2).contiguous(memory_format=torch.channels_last)
        return x4, nb, (1, 1, x.shape[2]), lambda o, sp: o.squeeze(2)
    if op.nd == 2:
        return (x.contiguous(memory_format=torch.channels_last), nb, (1, x.shape[2], x.shape[3]),
                lambda o, sp: o)
    return (x.contiguous(memory_format=torch.channels_last_3d), nb, tuple(x.shape[2:]), lambda o, sp: o)


def _alloc_out(op, nb, out_sp, dtype, device):
    if op.nd == 0:
        return torch.empty((nb, op.out_channels), dtype=dtype, device=device)
    if op.nd == 3:
        return torch.empty((nb, op.out_channels) + tuple(out_sp), dtype=dtype, device=device,
                           memory_format=torch.channels_last_3d)
    return torch.empty((nb, op.out_channels, out_sp[1], out_sp[2]), dtype=dtype, device=device,
                       memory_format=torch.channels_last)


def _sign_to_int8_cl(s, op):
    """explicit +/-1 sign tensor in logical layout -> int8 channels-last physical buffer."""
    phys, _, _, _ = _to_channels_last(s.to(torch.int8), op)
    return phys


def _weight_geom(op):
    """BtxGeom carrying what the sampled-weight tile layout depends on (N, K = taps*C/groups, groups)"""
    g = _lib.Geom()
    g.NB, g.D, g.H, g.W = 1, max(op.kernel[0], 1), max(op.kernel[1], 1), max(op.kernel[2], 1)
    g.C, g.N = op.in_channels, op.out_channels
    g.KD, g.KH, g.KW = op.kernel
    g.sd = g.sh = g.sw = 1
    g.dd = g.dh = g.dw = 1
    g.groups = op.groups
    return g


def sample_weights(items, seed, sample_idx, prec, device, sample_dev=None, lanes=1, bufs=None, skip_mu=False):
    """btx_sample_weights[_lanes]: ONE launch samples the weights of every layer in `items` for MC sample `sample_idx`
    (lanes > 1: for the samples sample_idx .. sample_idx + lanes - 1, or the `lanes` indices in `sample_dev`).
    items = [(kind, op, mu_p, rho_p, layer_id[, src_kw, src_c])] with GEMM-major f32 mu/rho as handed to contract_hip —
    or, with (src_kw, src_c), the UNPADDED weights of the padded geometry `op` (BtxSampleItem.src_KW/src_C); returns
    the list of uint8 tile buffers, to be passed as contract_hip(..., sampled_w=buf).  `bufs`: buffers of an earlier
    call with the same items to fill again; with skip_mu the mean tiles they hold are kept (BTX_SAMPLE_SKIP_MU: the
    caller vouches that mu has not changed since)."""
    L = _lib.lib()
    if not items:
        return []
    prec_c = _lib.PREC_CODE[prec]
    arr = (_lib.SampleItem * len(items))()
    geoms, outs = [], []
    for i, (kind, op, mu_p, rho_p, layer_id, *src) in enumerate(items):
        g = _weight_geom(op)
        geoms.append(g)
        nbytes = L.btx_sampled_w_bytes_lanes(ctypes.byref(g), kind, prec_c, int(lanes))
        if nbytes == 0:
            raise _lib.BtxError("btx_sampled_w_bytes: unsupported geometry")
        buf = bufs[i] if bufs is not None else torch.empty(nbytes, dtype=torch.uint8, device=device)
        if buf.numel() != nbytes:
            raise _lib.BtxError("sample_weights: buffer %d does not match the item (bytes %d != %d)" % (i, buf.numel(), nbytes))
        outs.append(buf)
        arr[i].geom = ctypes.pointer(g)
        arr[i].mu_w, arr[i].rho_w, arr[i].out = mu_p.data_ptr(), rho_p.data_ptr(), buf.data_ptr()
        arr[i].kind, arr[i].layer_id = kind, int(layer_id) & 0xFFFFFFFF
        if src:
            arr[i].src_KW, arr[i].src_C = int(src[0]), int(src[1])
    stream = torch.cuda.current_stream(device).cuda_stream
    r = _lib.Rng(int(seed), int(sample_idx) & 0xFFFFFFFF, 0, sample_dev.data_ptr() if sample_dev is not None else None)
    _lib.check(L.btx_sample_weights_lanes(arr, len(items), ctypes.byref(r), prec_c, stream, int(lanes),
                                          _lib.SAMPLE_SKIP_MU if skip_mu else 0))
    return outs


def contract_hip(kind, x, mu_p, rho_p, mu_b, rho_b, op, seed, sample_idx, layer_id, prec=None, noise=None,
                 extra_flags=0, out_dtype=None, epilogue=None, sampled_w=None, sample_dev=None, lanes=1, lane_batch=None,
                 self_sampling_weights=None):
    """One fused sample-and-contract forward on the GPU (btx_contract_fwd).  `mu_p`/`rho_p` are GEMM-major packed.
    `noise` (parity mode) = dict with optional eps_w (logical weight layout), eps_b, sign_in, sign_out.
    lanes > 1 (btx_contract_fwd_lanes): `lanes` MC samples in one launch.  x holds either the lanes' inputs back to
    back along the batch axis (lanes * lane_batch rows / images) or ONE input shared by all lanes (lane_batch rows);
    the output always holds the lanes back to back.  Sample indices: sample_idx + lane, or the `lanes` words of
    sample_dev.  self_sampling_weights: callable -> (mu_p, rho_p) a launch that samples for itself must read, for callers
    that pass `sampled_w` together with parameters the launch cannot sample from (the row-fused stem hands over its
    UNPADDED mu/rho when pre-sampled tiles exist); used by the per-lane fallback below, which drops the lane-batched tiles."""
    L = _lib.lib()
    if not x.is_cuda:
        raise _lib.BtxError("contract_hip needs a CUDA (ROCm) tensor")
    if x.dtype == torch.float32:
        act = _lib.ACT_F32
    elif x.dtype == torch.bfloat16:
        act = _lib.ACT_BF16
    else:
        raise _lib.BtxError("activations must be float32 or bfloat16, got %s" % x.dtype)
    prec = prec or _PRECISION
    prec_c = _lib.PREC_CODE[prec]
    xp, nb, spatial, restore = _to_channels_last(x, op)
    out_sp = op.out_spatial(spatial)
    if min(out_sp) <= 0:
        raise ValueError("output size is too small")
    lanes = int(lanes)
    nb_out, x_shared = nb, False
    if lanes > 1:
        if noise:
            raise _lib.BtxError("explicit noise tensors are single-sample (lanes == 1)")
        if lane_batch is None:
            raise _lib.BtxError("lanes > 1 needs lane_batch")
        rows = int(lane_batch) * (nb // x.shape[0] if (op.nd == 0 and x.dim() > 1 and x.shape[0] > 0) else 1)
        if nb == rows:
            x_shared, nb_out = True, rows * lanes
        elif nb != rows * lanes:
            raise _lib.BtxError("lanes=%d x lane_batch=%d does not match the input batch %d" % (lanes, lane_batch, x.shape[0]))
        nb = rows  # the geometry of ONE lane
        # btx_contract_fwd_lanes wants 16-byte lane strides.  A layer whose per-lane tensors are not (a 10-class head at an
        # odd batch: 10 * bs * 4 bytes) runs its lanes as single-sample launches planned like lanes (BTX_FLAG_CONCURRENT):
        # the same values, one launch per lane instead of one
        esz_o = (out_dtype or x.dtype).itemsize
        out_lane = nb * out_sp[0] * out_sp[1] * out_sp[2] * op.out_channels * esz_o
        x_lane = 0 if x_shared else (xp.numel() // lanes) * xp.element_size()
        pooled = epilogue is not None and epilogue.get("pool")
        if ((x_lane | out_lane) & 15) and not pooled:
            outs = []
            if sampled_w is not None and self_sampling_weights is not None:
                mu_p, rho_p = self_sampling_weights()  # the single-lane launches sample in registers, from these
            elif sampled_w is not None and (extra_flags & _lib.FLAG_ROWFUSE):
                raise _lib.BtxError("row-fused launch with pre-sampled tiles and unaligned lane strides: the per-lane fallback "
                                    "needs the padded parameters (self_sampling_weights)")
            xl = x.reshape((lanes, -1) + tuple(x.shape[1:])) if not x_shared else None
            res = epilogue.get("residual") if epilogue is not None else None
            for l in range(lanes):
                ep_l = epilogue
                if res is not None:
                    ep_l = dict(epilogue, residual=res.reshape((lanes, -1) + tuple(res.shape[1:]))[l])
                with concurrent_plan():
                    outs.append(contract_hip(kind, x if x_shared else xl[l], mu_p, rho_p, mu_b, rho_b, op, seed,
                                             int(sample_idx) + l, layer_id, prec=prec, extra_flags=extra_flags,
                                             out_dtype=out_dtype, epilogue=ep_l,
                                             sample_dev=sample_dev[l:l + 1] if sample_dev is not None else None))
            return torch.cat(outs, 0)
    flags = (_lib.FLAG_TRANSPOSED if op.transposed else 0) | extra_flags | (_lib.FLAG_CONCURRENT if _CONCURRENT else 0)
    if lanes > 1:
        flags |= lanes << _lib.FLAG_LANES_SHIFT
    rev = 0
    if _ALT_ORDER:
        _ORDER_TOGGLE[0] ^= 1
        rev = _lib.FLAG_REVERSE if _ORDER_TOGGLE[0] else 0
    if out_dtype is not None and out_dtype != x.dtype:
        flags |= _lib.FLAG_OUT_BF16 if out_dtype == torch.bfloat16 else _lib.FLAG_OUT_F32
    # the geometry struct and the workspace size depend on shapes only: built once per (op, batch, extent, modes)
    gkey = (id(op), nb, spatial, kind, act, prec_c, flags)
    cached = _GEOM_CACHE.get(gkey)
    if cached is None or cached[2] is not op:
        g = _lib.Geom()
        g.NB, (g.D, g.H, g.W), g.C, g.N = nb, spatial, op.in_channels, op.out_channels
        g.KD, g.KH, g.KW = op.kernel
        g.sd, g.sh, g.sw = op.stride
        g.pd, g.ph, g.pw = op.padding
        g.dd, g.dh, g.dw = op.dilation
        g.od, g.oh, g.ow = op.output_padding
        g.groups = op.groups
        if len(_GEOM_CACHE) > 4096:
            _GEOM_CACHE.clear()
        cached = _GEOM_CACHE[gkey] = (g, L.btx_contract_workspace_bytes(ctypes.byref(g), kind, act, prec_c, flags), op)
    g, need = cached[0], cached[1]
    if epilogue is not None and epilogue.get("pool"):  # BtxEpilogue.pool: `out` is the max-pooled tensor
        hq, wq = ctypes.c_int32(0), ctypes.c_int32(0)
        if not L.btx_contract_pool_shape(ctypes.byref(g), act, prec_c, flags, ctypes.byref(hq), ctypes.byref(wq)):
            raise _lib.BtxError("the fused stem max-pool is not available for this geometry (contract_pool_ok)")
        out_sp = (1, hq.value, wq.value)
    out = _alloc_out(op, nb_out, out_sp, out_dtype or x.dtype, x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    ws = _workspace(x.device, need, stream) if need else None
    r = _lib.Rng(int(seed), int(sample_idx) & 0xFFFFFFFF, int(layer_id) & 0xFFFFFFFF,
                 sample_dev.data_ptr() if sample_dev is not None else None)
    keep = []
    nz = None
    if sampled_w is not None:
        nz = _lib.Noise()
        nz.sampled_w = sampled_w.data_ptr()
        keep.append(sampled_w)
    if noise:
        nz = nz or _lib.Noise()
        if noise.get("eps_w_packed") is not None:  # already in the GEMM-major order of `op` (dgrad_weights_hip)
            t = noise["eps_w_packed"]
            keep.append(t); nz.eps_w = t.data_ptr()
        elif noise.get("eps_w") is not None:
            t = pack_gemm_major(noise["eps_w"].to(device=x.device, dtype=torch.float32), op)
            keep.append(t); nz.eps_w = t.data_ptr()
        if noise.get("eps_b") is not None:
            t = noise["eps_b"].to(device=x.device, dtype=torch.float32).contiguous()
            keep.append(t); nz.eps_b = t.data_ptr()
        if noise.get("sign_in") is not None:
            t = _sign_to_int8_cl(noise["sign_in"].to(x.device), op)
            keep.append(t); nz.sign_in = t.data_ptr()
        if noise.get("sign_out") is not None:
            so = noise["sign_out"].to(x.device)
            if op.nd == 0:
                so = so.reshape(-1, op.out_channels)
            t = _sign_to_int8_cl(so, op)
            keep.append(t); nz.sign_out = t.data_ptr()
    ev0 = None
    if _LAUNCH_LOG is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(x.device))
    ep = None
    if epilogue is not None:  # dict(scale, shift, residual, relu): eval-BN / residual / ReLU folded into the store
        ep = _lib.Epilogue()
        for name in ("scale", "shift"):
            t = epilogue.get(name)
            if t is not None:
                if t.dtype != torch.float32 or t.numel() != op.out_channels or not t.is_contiguous():
                    raise ValueError("epilogue %s must be a contiguous float32 [out_channels] tensor" % name)
                keep.append(t)
                setattr(ep, name, t.data_ptr())
        res = epilogue.get("residual")
        if res is not None:
            if op.nd == 0:
                res = res.reshape(-1, op.out_channels)
            rp, _, _, _ = _to_channels_last(res, OpDesc(op.nd, op.out_channels, op.out_channels))
            if rp.dtype != out.dtype or rp.numel() != out.numel():
                raise ValueError("epilogue residual must match the output shape and dtype")
            keep.append(rp)
            ep.residual = rp.data_ptr()
        ep.relu = 1 if epilogue.get("relu") else 0
        ep.pool = 1 if epilogue.get("pool") else 0
    args = (kind, ctypes.byref(g), xp.data_ptr(), mu_p.data_ptr(), rho_p.data_ptr(),
            mu_b.data_ptr() if mu_b is not None else None, rho_b.data_ptr() if rho_b is not None else None,
            out.data_ptr(), ctypes.byref(r), ctypes.byref(nz) if nz is not None else None,
            act, prec_c, flags | rev, ws.data_ptr() if ws is not None else None,
            ws.numel() if ws is not None else 0, stream, ctypes.byref(ep) if ep is not None else None)
    if lanes > 1:
        ln = _lib.Lanes()
        ln.n = lanes
        ln.x_stride = 0 if x_shared else (xp.numel() // lanes) * xp.element_size()
        ln.out_stride = (out.numel() // lanes) * out.element_size()
        ln.res_stride = ln.out_stride
        rc = L.btx_contract_fwd_lanes(*args, ctypes.byref(ln))
    else:
        rc = L.btx_contract_fwd_ex(*args)
    _lib.check(rc)
    if ev0 is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(torch.cuda.current_stream(x.device))
        m_rows = nb_out * out_sp[0] * out_sp[1] * out_sp[2]
        k_red = op.kernel[0] * op.kernel[1] * op.kernel[2] * (op.in_channels // op.groups)
        flops = 2.0 * m_rows * op.out_channels * k_red * (2 if kind == _lib.KIND_FLIPOUT else 1)
        tag = "%s/%s/%s k%dx%dx%d cin%d cout%d M%d" % ("flipout" if kind else "reparam", prec,
                                                     "bf16" if act == _lib.ACT_BF16 else "f32", op.kernel[0],
                                                     op.kernel[1], op.kernel[2], op.in_channels, op.out_channels, m_rows)
        _LAUNCH_LOG.append((tag, flops, ev0, ev1))
    if op.nd == 0 and x_shared:
        # a Linear layer that reads ONE input for all lanes (the first layer of an MLP): the output stacks the lanes along the
        # leading axis — [lanes * d0, *d1.., N] — where `restore` would fold it back into the input's own leading shape
        res = out.reshape((lanes * x.shape[0],) + tuple(x.shape[1:-1]) + (op.out_channels,))
    else:
        res = restore(out, out_sp)
    return res.contiguous() if (_OUT_LAYOUT == "contiguous" and op.nd > 0) else res


WGRAD_ATOMICS = False  # A/B and tests: accumulate the weight gradient with f32 atomics instead of chunk slabs


def wgrad_hip(kind, x, dy, op, seed, sample_idx, layer_id, w_shape, signs=None, swap=False, bias=False, rowfuse=None,
              _flags=0, raw=False, sample_dev=None, rho_flat=None):
    """btx_contract_wgrad: (dW_mu, dW_delta | None, db_mu | None, db_delta | None) in the layer's LOGICAL weight layout
    (f32).  `op` is a plain (non-transposed) contraction; `signs` = (sign_in, sign_out) logical +/-1 tensors for layers
    whose forward ran on padded layouts, else the forward's hashed signs are regenerated.  `rowfuse` = the layer's
    rowfuse_plan(): the gradient is taken on the row-fused geometry the forward ran on (hashed signs, a kernel row as the
    channel axis) and un-padded here.  `rho_flat` (with raw=True): the layer's rho in GEMM-major order — a fifth result, drho =
    dW * eps * sigmoid(rho) (dW = dW_delta, whose buffer then HOLDS drho, for Flipout; dW_mu for Reparameterization), formed by the
    slab reduction launch itself (btx_contract_wgrad_ws) or, on the atomics path, by btx_rho_grad."""
    L = _lib.lib()
    if op.transposed:
        raise _lib.BtxError("wgrad_hip wants the plain-convolution geometry (exchange x and dy for transposed layers)")
    if rowfuse is not None:
        plan, fop = rowfuse, rowfuse["op"]
        xin = rowfuse_input(x, plan, dy.dtype)
        fo = fop.out_spatial((1, plan["Hp"], plan["Wp"]))
        if (fo[1], fo[2]) != tuple(dy.shape[2:]):  # the forward computed (and dropped) extra rows / columns: their dy is 0
            dy = F.pad(dy, (0, fo[2] - dy.shape[3], 0, fo[1] - dy.shape[2]))
        kh, kwp, cp, kw, cin = fop.kernel[1], plan["kwp"], plan["cp"], plan["kw"], plan["cin"]
        dwm, dwd, dbm, dbd = wgrad_hip(kind, xin, dy, fop, seed, sample_idx, layer_id, (fop.out_channels, cp, kh, kwp),
                                       bias=bias, _flags=_lib.FLAG_ROWFUSE, sample_dev=sample_dev)
        un = lambda t: t[:, :cin, :, :kw].contiguous() if t is not None else None  # noqa: E731
        return un(dwm), un(dwd), dbm, dbd
    xp, nb, spatial, _ = _to_channels_last(x, op)
    yop = OpDesc(op.nd, op.out_channels, op.out_channels)
    dy2 = dy.reshape(-1, op.out_channels) if op.nd == 0 else dy
    dyp, _, _, _ = _to_channels_last(dy2, yop)
    if xp.dtype != dyp.dtype:
        dyp = dyp.to(xp.dtype)
    act = _lib.ACT_BF16 if xp.dtype == torch.bfloat16 else _lib.ACT_F32
    if xp.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.BtxError("activations must be float32 or bfloat16, got %s" % xp.dtype)
    g = _lib.Geom()
    g.NB, (g.D, g.H, g.W), g.C, g.N = nb, spatial, op.in_channels, op.out_channels
    g.KD, g.KH, g.KW = op.kernel
    g.sd, g.sh, g.sw = op.stride
    g.pd, g.ph, g.pw = op.padding
    g.dd, g.dh, g.dw = op.dilation
    g.groups = op.groups
    n, kred = op.out_channels, op.kernel[0] * op.kernel[1] * op.kernel[2] * (op.in_channels // op.groups)
    dev = x.device
    dwm = torch.empty(n * kred, dtype=torch.float32, device=dev)
    flip = kind == _lib.KIND_FLIPOUT
    dwd = torch.empty(n * kred, dtype=torch.float32, device=dev) if flip else None
    dbm = torch.empty(n, dtype=torch.float32, device=dev) if bias else None
    dbd = torch.empty(n, dtype=torch.float32, device=dev) if (bias and flip) else None
    nz, keep = None, []
    if signs is not None and flip:
        nz = _lib.Noise()
        si = _sign_to_int8_cl(signs[0].reshape(x.shape), op)
        so = _sign_to_int8_cl(signs[1].reshape(dy2.shape), yop)
        keep += [si, so]
        nz.sign_in, nz.sign_out = si.data_ptr(), so.data_ptr()
    r = _lib.Rng(int(seed), int(sample_idx) & 0xFFFFFFFF, int(layer_id) & 0xFFFFFFFF,
                 sample_dev.data_ptr() if sample_dev is not None else None)
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    flags = (_lib.FLAG_SWAP_SIGNS if swap else 0) | _flags
    # the pixel chunks' partial sums go to slabs that a second launch adds in chunk order (deterministic; f32 atomics from 2048
    # workgroups cost more than the slab round trip — profiles/r05_experiments.txt E13).  WGRAD_ATOMICS: the round 2-5 path.
    wsb = 0 if WGRAD_ATOMICS else int(L.btx_wgrad_workspace_bytes(kind, ctypes.byref(g), act, flags))
    drho = None
    if rho_flat is not None:
        if not raw:
            raise _lib.BtxError("wgrad_hip: rho_flat goes with raw=True (GEMM-major buffers)")
        if rho_flat.dtype != torch.float32 or not rho_flat.is_contiguous() or rho_flat.numel() != n * kred:
            raise _lib.BtxError("wgrad_hip: rho_flat must be the contiguous f32 GEMM-major rho of the layer")
        drho = dwd if flip else torch.empty_like(dwm)
    if wsb > 0:
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.check(L.btx_contract_wgrad_ws(kind, ctypes.byref(g), xp.data_ptr(), dyp.data_ptr(), dwm.data_ptr(), ptr(dwd),
                                           ptr(dbm), ptr(dbd), ctypes.byref(r), ctypes.byref(nz) if nz is not None else None,
                                           act, flags, ws.data_ptr(), wsb, ptr(rho_flat), ptr(drho),
                                           torch.cuda.current_stream(dev).cuda_stream))
    else:
        _lib.check(L.btx_contract_wgrad(kind, ctypes.byref(g), xp.data_ptr(), dyp.data_ptr(), dwm.data_ptr(), ptr(dwd),
                                        ptr(dbm), ptr(dbd), ctypes.byref(r), ctypes.byref(nz) if nz is not None else None,
                                        act, flags, torch.cuda.current_stream(dev).cuda_stream))
        if rho_flat is not None:
            rho_grad_hip(dwd if flip else dwm, rho_flat, seed, sample_idx, layer_id, _lib.STREAM_EPS_W, out=drho, sample_dev=sample_dev)
    if raw:  # the flat GEMM-major buffers as the kernel wrote them (autograd: strided logical views, no unpack copies)
        return (dwm, dwd, dbm, dbd) if rho_flat is None else (dwm, dwd, dbm, dbd, drho)
    un = lambda t: unpack_gemm_major(t, w_shape, op) if t is not None else None  # noqa: E731
    return un(dwm), un(dwd), dbm, dbd


def kl_hip(mu, rho, prior_mu, prior_sigma, prior_mu_t=None, prior_sigma_t=None, out=None, accumulate=False):
    """mean Gaussian KL of one parameter tensor on the GPU (btx_kl_gauss) -> 0-d f32 tensor."""
    L = _lib.lib()
    mu_c, rho_c = mu.detach().contiguous(), rho.detach().contiguous()
    if mu_c.dtype != torch.float32 or rho_c.dtype != torch.float32:
        raise _lib.BtxError("KL kernel needs float32 parameters")
    n = mu_c.numel()
    stream = torch.cuda.current_stream(mu.device).cuda_stream
    ws = _workspace(mu.device, L.btx_kl_workspace_bytes(n), stream)
    if out is None:
        out = torch.empty((), dtype=torch.float32, device=mu.device)
    pm = prior_mu_t.detach().contiguous() if prior_mu_t is not None else None
    ps = prior_sigma_t.detach().contiguous() if prior_sigma_t is not None else None
    rc = L.btx_kl_gauss(mu_c.data_ptr(), rho_c.data_ptr(), n, pm.data_ptr() if pm is not None else None,
                        ps.data_ptr() if ps is not None else None, float(prior_mu), float(prior_sigma),
                        out.data_ptr(), _lib.FLAG_KL_ACCUM if accumulate else 0, ws.data_ptr(), ws.numel(), stream)
    _lib.check(rc)
    return out


def _kl_items(entries, grads=None):
    """entries = [(mu, rho, prior_mu, prior_sigma, prior_mu_t|None, prior_sigma_t|None)] of same-order f32 tensors"""
    arr = (_lib.KlItem * len(entries))()
    for i, (mu, rho, pm, ps, pmt, pst) in enumerate(entries):
        arr[i].mu, arr[i].rho, arr[i].n = mu.data_ptr(), rho.data_ptr(), mu.numel()
        arr[i].prior_mu, arr[i].prior_sigma = float(pm), float(ps)
        arr[i].prior_mu_t = pmt.data_ptr() if pmt is not None else None
        arr[i].prior_sigma_t = pst.data_ptr() if pst is not None else None
        if grads is not None:
            arr[i].dmu, arr[i].drho = grads[i][0].data_ptr(), grads[i][1].data_ptr()
    return arr


def kl_model_hip(entries):
    """sum over tensors of the mean Gaussian KL, ONE launch + one final reduce (btx_kl_gauss_model) -> 0-d f32"""
    L = _lib.lib()
    dev = entries[0][0].device
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = _workspace(dev, L.btx_kl_model_workspace_bytes(len(entries)), stream)
    out = torch.empty((), dtype=torch.float32, device=dev)
    _lib.check(L.btx_kl_gauss_model(_kl_items(entries), len(entries), out.data_ptr(), ws.data_ptr(), ws.numel(), stream))
    return out


def kl_model_bwd_hip(entries, grads, grad_out):
    """d(kl)/d(mu, rho) of every tensor into `grads` [(dmu, drho)] (same element order), scaled by the device scalar
    `grad_out` (btx_kl_gauss_model_bwd)"""
    L = _lib.lib()
    dev = entries[0][0].device
    g = grad_out.detach().reshape(()).to(device=dev, dtype=torch.float32).contiguous()
    _lib.check(L.btx_kl_gauss_model_bwd(_kl_items(entries, grads), len(entries), g.data_ptr(),
                                        torch.cuda.current_stream(dev).cuda_stream))


def rowfuse_plan(op, x_shape):
    """Geometry of the row-fused execution of a small-C 2-D stem conv (BTX_FLAG_ROWFUSE), or None.  Decided by the
    layer geometry ONLY (never by dtypes), because the BTX-RNG index space of the layer follows this layout.
    The input is zero-padded to [N][Hp][Wp][cp] with the conv padding materialised and the kernel row padded to kwp
    taps; one kernel row = kwp*cp contiguous elements = a whole number of K-stages of the LDS-DMA kernel.
    Even stride_w: cp=4 (8-byte pixels stay 16-byte aligned), else cp=8."""
    if op.nd != 2 or op.transposed or op.groups != 1 or op.in_channels > 4 or op.dilation != (1, 1, 1):
        return None
    kh, kw = op.kernel[1], op.kernel[2]
    sh, sw = op.stride[1], op.stride[2]
    if kw > 8:
        return None
    if sw % 2 == 0:
        cp, kwp = 4, 8
    else:
        cp, kwp = 8, (4 if kw <= 4 else 8)
    ph, pw = op.padding[1], op.padding[2]
    H, W = x_shape[2], x_shape[3]
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    Wp = max((Wo - 1) * sw + kwp, W + 2 * pw)
    Wp += Wp % 2          # row pitch a multiple of 16 bytes for cp=4 bf16
    Hp = H + 2 * ph
    fop = OpDesc(2, cp, op.out_channels, (kh, kwp), (sh, sw), 0, 1, 1)
    return dict(op=fop, Hp=Hp, Wp=Wp, ph=ph, pw=pw, Ho=Ho, Wo=Wo, kw=kw, kwp=kwp, cp=cp, cin=op.in_channels)


def contract_pool_ok(op, nb, spatial, act_dtype, prec, extra_flags=0):
    """True when btx_contract_fwd_ex takes BtxEpilogue.pool = 1 (the ResNet stem's MaxPool2d(3, 2, 1) folded into the
    store of the row-fused stem contraction) for this geometry."""
    L = _lib.lib()
    g = _lib.Geom()
    g.NB, (g.D, g.H, g.W), g.C, g.N = nb, spatial, op.in_channels, op.out_channels
    g.KD, g.KH, g.KW = op.kernel
    g.sd, g.sh, g.sw = op.stride
    g.pd, g.ph, g.pw = op.padding
    g.dd, g.dh, g.dw = op.dilation
    g.od, g.oh, g.ow = op.output_padding
    g.groups = op.groups
    act = _lib.ACT_BF16 if act_dtype == torch.bfloat16 else _lib.ACT_F32
    prec_c = _lib.PREC_CODE[prec]
    return bool(L.btx_contract_pool_shape(ctypes.byref(g), act, prec_c, extra_flags, None, None))


def rowfuse_input(x, plan, out_dtype=None):
    """logical [N,C,H,W] -> zero-padded logical [N,cp,Hp,Wp] stored channels-last (in `out_dtype`).  CUDA tensors:
    one pass of btx_rowfuse_pack; CPU tensors: F.pad."""
    n, c, h, w = x.shape
    out_dtype = out_dtype or x.dtype
    if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and out_dtype in (torch.float32, torch.bfloat16):
        L = _lib.lib()
        out = torch.empty((n, plan["cp"], plan["Hp"], plan["Wp"]), dtype=out_dtype, device=x.device,
                          memory_format=torch.channels_last)
        st = (ctypes.c_int64 * 4)(*x.stride())
        code = lambda dt: _lib.ACT_BF16 if dt == torch.bfloat16 else _lib.ACT_F32
        _lib.check(L.btx_rowfuse_pack(x.data_ptr(), code(x.dtype), st, n, c, h, w, out.data_ptr(), code(out_dtype),
                                      plan["Hp"], plan["Wp"], plan["cp"], plan["ph"], plan["pw"],
                                      torch.cuda.current_stream(x.device).cuda_stream))
        return out
    x = x.to(out_dtype)
    xp = F.pad(x.permute(0, 2, 3, 1), (0, plan["cp"] - c, plan["pw"], plan["Wp"] - w - plan["pw"], plan["ph"],
                                       plan["Hp"] - h - plan["ph"]))
    return xp.permute(0, 3, 1, 2)


def rowfuse_weights(mu_p, rho_p, plan):
    """GEMM-major [Cout,KH,KW,C] -> [Cout,KH,kwp,cp]; the padded taps meet real pixels, so they must contribute
    exactly nothing: mu = 0 and rho = -1e30 (softplus -> 0)."""
    kw, c = plan["kw"], plan["cin"]
    mu_f = F.pad(mu_p, (0, plan["cp"] - c, 0, plan["kwp"] - kw))
    rho_f = F.pad(rho_p, (0, plan["cp"] - c, 0, plan["kwp"] - kw), value=-1e30)
    return mu_f, rho_f


def pad_channels(x, op, extra):
    """zero-pad the channel axis of a logical [N,C,*sp] (or [*,C]) tensor by `extra`, result channels-last"""
    nd = op.nd
    if nd == 0:
        return F.pad(x, (0, extra))
    perm = (0,) + tuple(range(2, 2 + nd)) + (1,)
    xp = F.pad(x.permute(perm), (0, extra))  # contiguous [N,*sp,C+extra]
    return xp.permute((0, nd + 1) + tuple(range(1, nd + 1)))


def fill_eps_hip(n, device, seed, sample_idx, layer_id, rng_stream, sample_dev=None):
    """BTX-RNG v1 eps for a flat index space of n elements, as a flat f32 CUDA tensor.  sample_dev: the sample index lives in
    that device word (captured steps) and sample_idx is ignored."""
    L = _lib.lib()
    out = torch.empty(int(n), dtype=torch.float32, device=device)
    r = _lib.Rng(int(seed), int(sample_idx) & 0xFFFFFFFF, int(layer_id) & 0xFFFFFFFF,
                 sample_dev.data_ptr() if sample_dev is not None else None)
    _lib.check(L.btx_fill_eps(out.data_ptr(), out.numel(), ctypes.byref(r), rng_stream,
                              torch.cuda.current_stream(out.device).cuda_stream))
    return out


def rho_grad_hip(dw_flat, rho_flat, seed, sample_idx, layer_id, rng_stream, out=None, sample_dev=None):
    """btx_rho_grad: drho = dw * eps * sigmoid(rho) over flat f32 tensors in the same (GEMM-major) element order, eps
    regenerated inside the kernel.  `out` may be dw_flat itself."""
    L = _lib.lib()
    if out is None:
        out = torch.empty_like(dw_flat)
    r = _lib.Rng(int(seed), int(sample_idx) & 0xFFFFFFFF, int(layer_id) & 0xFFFFFFFF,
                 sample_dev.data_ptr() if sample_dev is not None else None)
    _lib.check(L.btx_rho_grad(dw_flat.data_ptr(), rho_flat.data_ptr(), out.data_ptr(), dw_flat.numel(), ctypes.byref(r),
                              rng_stream, torch.cuda.current_stream(dw_flat.device).cuda_stream))
    return out


def dgrad_weights_hip(mu_p, rho_p, n, taps, c, flip, seed, sample_idx, layer_id, sample_dev=None):
    """btx_dgrad_weights: (mu, rho, eps) of the data gradient's transposed geometry, GEMM-major [c][taps][n] f32, from the layer's
    own GEMM-major parameters [n][taps][c] in ONE launch (eps = the forward's draw, regenerated)."""
    L = _lib.lib()
    dev = mu_p.device
    outs = [torch.empty(int(c) * int(taps) * int(n), dtype=torch.float32, device=dev) for _ in range(3)]
    r = _lib.Rng(int(seed), int(sample_idx) & 0xFFFFFFFF, int(layer_id) & 0xFFFFFFFF,
                 sample_dev.data_ptr() if sample_dev is not None else None)
    _lib.check(L.btx_dgrad_weights(mu_p.data_ptr(), rho_p.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                   int(n), int(taps), int(c), 1 if flip else 0, ctypes.byref(r),
                                   torch.cuda.current_stream(dev).cuda_stream))
    return outs


def fill_sign_hip(n, device, seed, sample_idx, layer_id, rng_stream):
    L = _lib.lib()
    out = torch.empty(int(n), dtype=torch.int8, device=device)
    r = _lib.Rng(int(seed), int(sample_idx) & 0xFFFFFFFF, int(layer_id) & 0xFFFFFFFF)
    _lib.check(L.btx_fill_sign(out.data_ptr(), out.numel(), ctypes.byref(r), rng_stream,
                               torch.cuda.current_stream(out.device).cuda_stream))
    return out


def gemm_major_logical_view(buf, w_shape, op):
    """The logical-shape tensor (reference layout [out,in] / [Cout,Cin/g,*k] / [Cin,Cout/g,*k]) over a GEMM-major buffer
    `buf` (what gemm_major_view / pack_gemm_major produce): a strided VIEW of `buf` where one exists — the inverse of
    gemm_major_param — else (ConvTranspose with groups > 1) an unpacked copy."""
    nd = op.nd
    if nd == 0:
        return buf.reshape(w_shape)
    k = tuple(w_shape[2:])
    if not op.transposed:
        return buf.reshape((w_shape[0],) + k + (w_shape[1],)).permute((0, nd + 1) + tuple(range(1, nd + 1)))
    if op.groups == 1:
        return buf.reshape((w_shape[1],) + k + (w_shape[0],)).permute((nd + 1, 0) + tuple(range(1, nd + 1)))
    return unpack_gemm_major(buf.reshape(-1), w_shape, op)


def unpack_gemm_major(flat, w_shape, op):
    """inverse of pack_gemm_major for a flat [N*taps*Cg] tensor -> logical parameter layout."""
    if op.nd == 0:
        return flat.reshape(w_shape)
    k = tuple(w_shape[2:])
    if not op.transposed:
        t = flat.reshape((w_shape[0],) + k + (w_shape[1],))
        perm = (0, op.nd + 1) + tuple(range(1, op.nd + 1))
        return t.permute(perm).contiguous()
    g = op.groups
    cin, ng = w_shape[0], w_shape[1]
    t = flat.reshape(g, ng, -1, cin // g).permute(0, 3, 1, 2).contiguous()  # [g, Cg, Ng, T]
    return t.reshape((cin, ng) + k)


# --------------------------------------------------------------------------------------------------------------
# ATen route for CPU tensors (BASELINE.json config 0) and for autograd-on-request (`backend="torch"`).
# Same op chain and the same torch-generator draw order as the reference methods cited in each layer class.
# --------------------------------------------------------------------------------------------------------------
_CONV = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}
_CONVT = {1: F.conv_transpose1d, 2: F.conv_transpose2d, 3: F.conv_transpose3d}


def contract_aten(x, w, b, op):
    nd = op.nd
    if nd == 0:
        return F.linear(x, w, b)
    st, pd, dl, opd = op.stride[3 - nd:], op.padding[3 - nd:], op.dilation[3 - nd:], op.output_padding[3 - nd:]
    if op.transposed:
        return _CONVT[nd](x, w, b, st, pd, opd, op.groups, dl)
    return _CONV[nd](x, w, b, st, pd, dl, op.groups)


def plain_layout(t):
    """row-major copy with DEFAULT strides (`.contiguous()` is not enough: a [Cout,Cin,1,1] view with
    channels-last-looking strides already counts as contiguous and would steer ATen's conv to channels_last)."""
    return t.clone(memory_format=torch.contiguous_format)


def softplus_naive(rho):
    return torch.log1p(torch.exp(rho))


def kl_aten(mu_q, sigma_q, mu_p, sigma_p):
    kl = torch.log(sigma_p) - torch.log(sigma_q) + (sigma_q ** 2 + (mu_q - mu_p) ** 2) / (2 * (sigma_p ** 2)) - 0.5
    return kl.mean()


def maxpool2d_hip(x, kernel, stride, padding):
    """torch.nn.functional.max_pool2d for channels-last CUDA tensors through btx_maxpool2d_cl (C % 8 == 0)"""
    L = _lib.lib()
    n, c, h, w = x.shape
    xp = x.contiguous(memory_format=torch.channels_last)
    ho, wo = (h + 2 * padding - kernel) // stride + 1, (w + 2 * padding - kernel) // stride + 1
    out = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _lib.check(L.btx_maxpool2d_cl(xp.data_ptr(), out.data_ptr(), _lib.ACT_BF16 if x.dtype == torch.bfloat16 else _lib.ACT_F32,
                                  n, h, w, c, kernel, stride, padding, torch.cuda.current_stream(x.device).cuda_stream))
    return out


def maxpool2d_train_hip(x, kernel, stride, padding):
    """(y, idx) of btx_maxpool2d_cl_train: max_pool2d of a channels-last CUDA tensor and, per output element, the window position of
    its maximum (uint8 [N][Ho][Wo][C]) for maxpool2d_bwd_hip"""
    L = _lib.lib()
    n, c, h, w = x.shape
    ho, wo = (h + 2 * padding - kernel) // stride + 1, (w + 2 * padding - kernel) // stride + 1
    out = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    idx = torch.empty(n * ho * wo * c, dtype=torch.uint8, device=x.device)
    _lib.check(L.btx_maxpool2d_cl_train(x.data_ptr(), out.data_ptr(), idx.data_ptr(), _lib.ACT_BF16 if x.dtype == torch.bfloat16 else _lib.ACT_F32,
                                        n, h, w, c, kernel, stride, padding, torch.cuda.current_stream(x.device).cuda_stream))
    return out, idx


def maxpool2d_bwd_hip(dy, idx, x_shape, kernel, stride, padding):
    """btx_maxpool2d_cl_bwd: the gradient of maxpool2d_train_hip's input (channels-last), dy channels-last"""
    L = _lib.lib()
    n, c, h, w = x_shape
    dx = torch.empty((n, c, h, w), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    _lib.check(L.btx_maxpool2d_cl_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), _lib.ACT_BF16 if dy.dtype == torch.bfloat16 else _lib.ACT_F32,
                                      n, h, w, c, kernel, stride, padding, torch.cuda.current_stream(dy.device).cuda_stream))
    return dx


def avgpool_global_hip(x):
    """adaptive_avg_pool2d(x, 1).flatten(1) for channels-last CUDA tensors through btx_avgpool_global_cl (C % 8 == 0)"""
    L = _lib.lib()
    n, c, h, w = x.shape
    xp = x.contiguous(memory_format=torch.channels_last)
    out = torch.empty((n, c), dtype=x.dtype, device=x.device)
    _lib.check(L.btx_avgpool_global_cl(xp.data_ptr(), out.data_ptr(), _lib.ACT_BF16 if x.dtype == torch.bfloat16 else _lib.ACT_F32,
                                       n, h * w, c, torch.cuda.current_stream(x.device).cuda_stream))
    return out
