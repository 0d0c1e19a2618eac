"""Training path of the variational layers on the HIP backend (reference README.md:114-125:
`output = model(x); kl = get_kl_loss(model); loss = ce(output, y) + kl / batch_size; loss.backward()`).

BTX-RNG noise is a pure function of (seed, sample_idx, layer_id, element index), so nothing but (x, mu, rho) is saved for
the backward: eps and the Flipout signs are regenerated.

  forward        the fused HIP contraction (btx_contract_fwd), unchanged
  dx             ALSO btx_contract_fwd: the data gradient of  y = conv(x, mu) + s_out * conv(x * s_in, sigma*eps)  is
                 dx = convT(dy, mu) + s_in * convT(dy * s_out, sigma*eps)  — the same Flipout-shaped contraction on the
                 transposed geometry (stride-1 2-D convolutions: on the flipped kernel, which the tap-unrolled kernel
                 takes) with the two sign streams exchanged (BTX_FLAG_SWAP_SIGNS) and the forward's eps passed explicitly
  KL, dKL        btx_kl_gauss_model / btx_kl_gauss_model_bwd (one launch for the whole model)
  dmu, drho      btx_contract_wgrad: dW_mu = corr(x, dy) [and dW_delta = corr(x*s_in, dy*s_out) for Flipout] — a GEMM whose
                 reduction axis is the pixel axis, on the exact-f32 MFMA (csrc/btx_wgrad.hip); dmu = dW_mu and
                 drho = dW_delta * eps * sigmoid(rho) are elementwise follow-ups here.
"""
import ctypes

import torch

from . import _lib
from . import functional as BF
from . import rng as _rng


def fast_dgrad_ok(layer):
    """the data gradient's weight operands in ONE launch (btx_dgrad_weights): Linear layers and 2-D convolutions on plain layouts
    (no channel padding, no row-fused stem, groups == 1); at stride 1 the padding must not exceed the dilated kernel extent"""
    op = layer._op
    if layer._btx_cpad is not None or op.transposed or op.groups != 1:
        return False
    if op.nd == 0:
        return True
    if op.nd == 2 and op.in_channels > 4:
        if op.stride == (1, 1, 1):
            return all(d * (k - 1) - p >= 0 for d, k, p in zip(op.dilation[1:], op.kernel[1:], op.padding[1:]))
        return True  # strided: the transposed geometry on the un-flipped, channel-transposed kernel (the same one launch)
    return False


def _data_grad_hip(layer, dy, x_shape, nz, sample_idx, hashed_signs, mu, rho):
    """dx through libbtx: the contraction of dy with the (mu, sigma*eps) the FORWARD used (the tensors autograd saved) on
    the transposed geometry"""
    op = layer._op
    kind = _lib.KIND_FLIPOUT if layer._family == "flipout" else _lib.KIND_REPARAM
    nd = op.nd
    if fast_dgrad_ok(layer) and (kind == _lib.KIND_REPARAM or hashed_signs):
        mu_p, rho_p = BF.gemm_major_view(mu.detach(), op), BF.gemm_major_view(rho.detach(), op)
        if nd == 0:
            opT, taps, flip = BF.OpDesc(0, op.out_channels, op.in_channels), 1, False
        elif op.stride == (1, 1, 1):
            pad = tuple(d * (k - 1) - p for d, k, p in zip(op.dilation[1:], op.kernel[1:], op.padding[1:]))
            opT = BF.OpDesc(2, op.out_channels, op.in_channels, op.kernel[1:], 1, pad, op.dilation[1:], 1)
            taps, flip = op.kernel[1] * op.kernel[2], True
        else:
            # strided: dx = convT(dy, W); its GEMM-major operands [C][tap][N] are the channel transpose of the forward's [N][tap][C]
            outpad = tuple((i + 2 * p - d * (k - 1) - 1) % s_ for i, p, d, k, s_ in
                           zip(x_shape[2:], op.padding[1:], op.dilation[1:], op.kernel[1:], op.stride[1:]))
            opT = BF.OpDesc(2, op.out_channels, op.in_channels, op.kernel[1:], op.stride[1:], op.padding[1:], op.dilation[1:], 1,
                            transposed=True, output_padding=outpad)
            taps, flip = op.kernel[1] * op.kernel[2], False
        sdev = getattr(layer, "_btx_sample_dev", None)  # captured training step: the sample index is a device word
        w_mu, w_rho, w_eps = BF.dgrad_weights_hip(mu_p, rho_p, op.out_channels, taps, op.in_channels, flip, _rng.seed(),
                                                  sample_idx, layer._btx_layer_id, sample_dev=sdev)
        dx = BF.contract_hip(kind, dy, w_mu, w_rho, None, None, opT, _rng.seed(), sample_idx, layer._btx_layer_id,
                             prec=layer.precision, noise={"eps_w_packed": w_eps}, sample_dev=sdev,
                             extra_flags=_lib.FLAG_SWAP_SIGNS if kind == _lib.KIND_FLIPOUT else 0)
        return dx.reshape(x_shape) if nd == 0 else dx
    mu, rho, eps = mu.detach(), rho.detach(), nz["eps_w"]
    if nd == 0:
        opT = BF.OpDesc(0, op.out_channels, op.in_channels)
        w_mu, w_rho, w_eps = mu.t(), rho.t(), eps.t()
    elif op.transposed:
        # forward was a transposed convolution: its data gradient is the plain convolution with the same weight tensor
        opT = BF.OpDesc(nd, op.out_channels, op.in_channels, op.kernel[3 - nd:], op.stride[3 - nd:], op.padding[3 - nd:],
                        op.dilation[3 - nd:], op.groups)
        w_mu, w_rho, w_eps = mu, rho, eps
    elif nd == 2 and op.groups == 1 and op.stride == (1, 1, 1):
        # stride 1: convT(dy, W) == conv(dy, flip(W)^T) with padding d*(k-1) - p: a plain 3x3 again (tap-unrolled kernel)
        pad = tuple(d * (k - 1) - p for d, k, p in zip(op.dilation[1:], op.kernel[1:], op.padding[1:]))
        if min(pad) < 0:
            raise _lib.BtxError("data gradient: padding larger than the dilated kernel extent is not supported")
        opT = BF.OpDesc(2, op.out_channels, op.in_channels, op.kernel[1:], 1, pad, op.dilation[1:], 1)
        tr = lambda t: t.transpose(0, 1).flip(2, 3)  # noqa: E731
        w_mu, w_rho, w_eps = tr(mu), tr(rho), tr(eps)
    else:
        outpad = tuple((i + 2 * p - d * (k - 1) - 1) % s for i, p, d, k, s in
                       zip(x_shape[2:], op.padding[3 - nd:], op.dilation[3 - nd:], op.kernel[3 - nd:], op.stride[3 - nd:]))
        opT = BF.OpDesc(nd, op.out_channels, op.in_channels, op.kernel[3 - nd:], op.stride[3 - nd:], op.padding[3 - nd:],
                        op.dilation[3 - nd:], op.groups, transposed=True, output_padding=outpad)
        w_mu, w_rho, w_eps = mu, rho, eps
    noise = {"eps_w": w_eps}
    flags = 0
    if kind == _lib.KIND_FLIPOUT:
        if hashed_signs:
            flags = _lib.FLAG_SWAP_SIGNS
        else:  # padded forward layouts (C % 8 != 0, row-fused stems): the signs as tensors
            noise["sign_in"], noise["sign_out"] = nz["sign_out"], nz["sign_in"]
    dx = BF.contract_hip(kind, dy, BF.pack_gemm_major(w_mu.float(), opT), BF.pack_gemm_major(w_rho.float(), opT), None, None,
                         opT, _rng.seed(), sample_idx, layer._btx_layer_id, prec=layer.precision, noise=noise,
                         extra_flags=flags, sample_dev=getattr(layer, "_btx_sample_dev", None))
    return dx.reshape(x_shape) if nd == 0 else dx


class ContractFn(torch.autograd.Function):
    """y = layer(x) on the HIP backend with gradients w.r.t. x, mu, rho, mu_b, rho_b"""

    @staticmethod
    def forward(ctx, layer, sample_idx, x, mu, rho, mu_b, rho_b):
        with torch.no_grad():
            out = layer._forward_hip(x, sample_idx=sample_idx)
        ctx.layer, ctx.sample_idx = layer, sample_idx
        ctx.save_for_backward(x, mu, rho, rho_b)  # mu: the data gradient contracts with it (autograd's version check applies)
        return out

    @staticmethod
    def backward(ctx, dy):
        layer, s = ctx.layer, ctx.sample_idx
        x, mu_s, rho, rho_b = ctx.saved_tensors
        op = layer._op
        flip = layer._family == "flipout"
        dy = dy.contiguous() if op.nd == 0 else dy
        with torch.no_grad():
            plan = layer._rowfuse_plan(x) if op.nd == 2 else None
            padded = layer._btx_cpad is not None or plan is not None   # forward noise indices run over padded layouts
            # sign TENSORS are only needed where the kernels cannot regenerate the hashed signs: channel-padded layouts, and
            # the data gradient / transposed weight gradient of a row-fused stem
            stem_wgrad = plan is not None and not op.transposed
            need_signs = flip and ((padded and (ctx.needs_input_grad[2] or not stem_wgrad)) or
                                   (op.transposed and rho_b is not None))  # transposed layers: db_delta by torch
            w_shape = tuple(rho.shape)
            fused_w = False  # dmu / drho formed by btx_rho_grad on the kernel's own buffers (plain layouts)
            # the noise TENSORS (eps, signs) are only materialised where something still needs them: the data gradient (it
            # contracts with the transposed / flipped sigma*eps), padded / transposed layouts, bias gradients
            plain_w = not op.transposed and not padded
            fast_dx = plain_w and fast_dgrad_ok(layer)  # the data gradient regenerates eps in its own operand pass
            need_nz = ((ctx.needs_input_grad[2] and not fast_dx) or not plain_w or need_signs or
                       (rho_b is not None and (ctx.needs_input_grad[5] or ctx.needs_input_grad[6])))
            sdev = getattr(layer, "_btx_sample_dev", None)  # captured step: eps / signs follow the device word, not `s`
            nz = layer.materialize_noise(s, tuple(x.shape), tuple(dy.shape), x.dtype, signs=need_signs,
                                         sample_dev=sdev) if need_nz else {}
            dx = dmu = drho = dmu_b = drho_b = None
            kind = _lib.KIND_FLIPOUT if flip else _lib.KIND_REPARAM
            want_w = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
            want_b = rho_b is not None and (ctx.needs_input_grad[5] or ctx.needs_input_grad[6])
            if want_w or want_b:
                signs = (nz["sign_in"], nz["sign_out"]) if (flip and padded and "sign_in" in nz) else None
                if plan is not None and not op.transposed:
                    # row-fused stem: the gradient on the geometry the forward ran on (7 kernel rows x 32 elements instead of
                    # 49 taps x 3 channels of a 64-wide tile; the forward's hashed signs instead of sign tensors)
                    dW, dWd, db, dbd = BF.wgrad_hip(kind, x, dy, op, _rng.seed(), s, layer._btx_layer_id, w_shape,
                                                    bias=want_b, rowfuse=plan, sample_dev=sdev)
                elif not op.transposed and not padded:
                    # plain layouts: the kernel's GEMM-major buffers ARE the gradients (strided logical views, as the
                    # parameters themselves are stored), and drho = dW_delta * eps * sigmoid(rho) is one launch with eps
                    # regenerated in the kernel (btx_rho_grad) instead of fill_eps + unpack + sigmoid + two products
                    rho_f = BF.gemm_major_view(rho, op).reshape(-1) if want_w else None
                    if rho_f is not None and not rho_f.is_contiguous():
                        rho_f = rho_f.contiguous()
                    res = BF.wgrad_hip(kind, x, dy, op, _rng.seed(), s, layer._btx_layer_id, w_shape, bias=want_b, raw=True,
                                       sample_dev=sdev, rho_flat=rho_f)
                    dWf, dWdf, db, dbd = res[:4]
                    if want_w:
                        drho_f = res[4]  # formed by the weight gradient's own reduction launch (btx_contract_wgrad_ws)
                        dmu = BF.gemm_major_logical_view(dWf, w_shape, op)
                        drho = BF.gemm_major_logical_view(drho_f, w_shape, op)
                    fused_w = True
                elif not op.transposed:
                    dW, dWd, db, dbd = BF.wgrad_hip(kind, x, dy, op, _rng.seed(), s, layer._btx_layer_id, w_shape,
                                                    signs=signs, bias=want_b, sample_dev=sdev)
                else:
                    # y = convT(x, W): W is the weight of the plain convolution that maps y-space to x-space, so its
                    # gradient is corr(dy, x) on that geometry — x and dy (and the two sign streams) exchange roles
                    nd = op.nd
                    opc = BF.OpDesc(nd, op.out_channels, op.in_channels, op.kernel[3 - nd:], op.stride[3 - nd:],
                                    op.padding[3 - nd:], op.dilation[3 - nd:], op.groups)
                    sw = (signs[1], signs[0]) if signs is not None else None
                    dW, dWd, _, _ = BF.wgrad_hip(kind, dy, x, opc, _rng.seed(), s, layer._btx_layer_id, w_shape, signs=sw,
                                                 swap=True, sample_dev=sdev)
                    db = dbd = None
                    if want_b:
                        red = tuple(i for i in range(dy.dim()) if i != 1)
                        db = dy.float().sum(red)
                        dbd = (dy.float() * nz["sign_out"].reshape(dy.shape).float()).sum(red) if flip else None
                if want_w and not fused_w:
                    dmu = dW
                    drho = (dWd if flip else dW) * nz["eps_w"] * torch.sigmoid(rho.detach())
                if want_b:
                    dmu_b = db
                    drho_b = (dbd if flip else db) * nz["eps_b"] * torch.sigmoid(rho_b.detach())
            if ctx.needs_input_grad[2]:
                hashed = flip and not padded
                dx = _data_grad_hip(layer, dy, tuple(x.shape), nz, s, hashed, mu_s, rho)
                if dx.dtype != x.dtype:
                    dx = dx.to(x.dtype)
        return None, None, dx, dmu, drho, dmu_b, drho_b


class KlFn(torch.autograd.Function):
    """sum of the per-tensor mean KLs of `n` (mu, rho) pairs on the HIP backend, with gradients"""

    @staticmethod
    def forward(ctx, meta, *params):
        # meta: per pair (prior_mu, prior_sigma, prior_mu_t | None, prior_sigma_t | None, op | None)
        ctx.meta = meta
        ctx.save_for_backward(*params)
        return BF.kl_model_hip(_entries(meta, params))

    @staticmethod
    def backward(ctx, g):
        params = ctx.saved_tensors
        meta = ctx.meta
        entries = _entries(meta, params)
        grads, outs = [], []
        for i, (pm, ps, pmt, pst, op) in enumerate(meta):
            mu, rho = params[2 * i], params[2 * i + 1]
            if pmt is None and op is not None:
                # The kernel reads (mu, rho) in GEMM-major order (_entries) — a zero-copy view of the parameter's storage or,
                # for parameters that are not stored that way (ConvTranspose with groups > 1, a parameter re-assigned
                # contiguous), a packed copy — and writes the gradients in the SAME element order: into GEMM-major buffers,
                # handed back as logical-shape views of those buffers.
                gpm = torch.empty_like(BF.gemm_major_view(mu, op))
                gpr = torch.empty_like(gpm)
                grads.append((gpm, gpr))
                outs.append((gpm, gpr, tuple(mu.shape), op))
            else:
                gm = torch.empty(mu.shape, dtype=torch.float32, device=mu.device)
                gr = torch.empty(rho.shape, dtype=torch.float32, device=rho.device)
                grads.append((gm, gr))
                outs.append((gm, gr, None, None))
        BF.kl_model_bwd_hip(entries, grads, g)
        res = []
        for gm, gr, shape, op in outs:  # (views where the layout has one, unpacked copies — made NOW — where it does not)
            if op is not None:
                gm, gr = BF.gemm_major_logical_view(gm, shape, op), BF.gemm_major_logical_view(gr, shape, op)
            res += [gm, gr]
        return (None,) + tuple(res)


def _entries(meta, params):
    out = []
    for i, (pm, ps, pmt, pst, op) in enumerate(meta):
        mu, rho = params[2 * i].detach(), params[2 * i + 1].detach()
        if pmt is None and op is not None:
            out.append((BF.gemm_major_view(mu, op), BF.gemm_major_view(rho, op), pm, ps, None, None))
        else:  # element order must match the logical prior tensors
            out.append((mu.contiguous(), rho.contiguous(), pm, ps,
                        pmt.detach().contiguous() if pmt is not None else None,
                        pst.detach().contiguous() if pst is not None else None))
    return out


def kl_pairs(layer):
    """[(mu, rho, meta)] of one variational layer for KlFn / kl_model_hip"""
    mu, rho = layer._w()
    tens = not layer._priors_are_scalar()
    pairs = [(mu, rho, (layer.prior_mean, layer.prior_variance, layer.prior_weight_mu if tens else None,
                        layer.prior_weight_sigma if tens else None, layer._op))]
    if layer.mu_bias is not None:
        pairs.append((layer.mu_bias, layer.rho_bias, (layer.prior_mean, layer.prior_variance,
                                                      layer.prior_bias_mu if tens else None,
                                                      layer.prior_bias_sigma if tens else None, None)))
    return pairs


def kl_of_layers(layers):
    """get_kl_loss on the HIP backend: one launch for every tensor of every layer; differentiable when needed"""
    pairs = [pr for layer in layers for pr in kl_pairs(layer)]
    params = [t for mu, rho, _ in pairs for t in (mu, rho)]
    meta = tuple(m for _, _, m in pairs)
    if torch.is_grad_enabled() and any(t.requires_grad for t in params):
        return KlFn.apply(meta, *params)
    return BF.kl_model_hip(_entries(meta, params))


# --------------------------------------------------------------------------------------------------------------------
# training-mode BatchNorm on channels-last activations (csrc/btx_bn.hip, btx_bn_train_fwd / _bwd): the step either side of
# the variational convolutions in the reference's training loop (README.md:114-125 on models/deterministic/resnet_large.py)
# --------------------------------------------------------------------------------------------------------------------
def _act_code(dt):
    return _lib.ACT_BF16 if dt == torch.bfloat16 else _lib.ACT_F32


def bn_train_usable(bn, x):
    """True when nn.BatchNorm{1,2,3}d `bn` in training mode on `x` can take the HIP kernels: CUDA tensor, f32 / bf16,
    channels-last storage (or 2-D [M, C]), C % 8 == 0, parameters and running estimates of one dtype (f32 or bf16)."""
    if type(bn) not in (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d):
        return False  # nn.SyncBatchNorm reduces its statistics across ranks, Lazy* variants have no parameters yet: torch's own path
    if not (bn.training and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.dim() in (2, 4, 5)):
        return False
    C = x.shape[1]
    if C % 8 != 0 or C > 2048 or x.numel() == 0 or x.numel() // C < 2:
        return False
    if bn.momentum is None and bn.track_running_stats:
        return False  # cumulative moving average: torch's own path
    ts = [t for t in (bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                      bn.running_var if bn.track_running_stats else None) if t is not None]
    if ts and (any(t.dtype != ts[0].dtype for t in ts) or ts[0].dtype not in (torch.float32, torch.bfloat16)):
        return False
    if x.dim() == 2 and not x.is_contiguous():
        return False
    if x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
        return False
    if x.dim() == 5 and not x.is_contiguous(memory_format=torch.channels_last_3d):
        return False
    return True


class BatchNormTrainFn(torch.autograd.Function):
    """y = [relu](batch_norm(x) [+ residual]) with batch statistics; running_mean / running_var / num_batches_tracked updated in
    place (as F.batch_norm / nn.BatchNorm do).  The ReLU and the residual add of the reference's blocks
    (models/deterministic/resnet_large.py:46-62) ride in the normalisation's own launches (csrc/btx_bn.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, batches_tracked=None, residual=None, relu=False):
        L = _lib.lib()
        C = x.shape[1]
        M = x.numel() // C
        dev = x.device
        y = torch.empty_like(x)  # preserves the channels-last strides
        save_mean = torch.empty(C, dtype=torch.float32, device=dev)
        save_invstd = torch.empty(C, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = BF._workspace(dev, L.btx_bn_workspace_bytes(M, C), stream)
        ref = next((t for t in (weight, bias, running_mean, running_var) if t is not None), None)
        pdt = _act_code(ref.dtype) if ref is not None else _lib.ACT_F32
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        w = weight.detach().contiguous() if weight is not None else None
        b = bias.detach().contiguous() if bias is not None else None
        mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=dev) if relu else None  # one bit per element: y > 0
        fuse = None
        if relu or residual is not None:
            fuse = _lib.BnFuse(ptr(residual), 1 if relu else 0, ptr(mask), None)
        _lib.check(L.btx_bn_train_fwd(x.data_ptr(), y.data_ptr(), _act_code(x.dtype), M, C, ptr(w), ptr(b), ptr(running_mean),
                                      ptr(running_var), pdt, float(momentum if momentum is not None else 0.0), float(eps),
                                      save_mean.data_ptr(), save_invstd.data_ptr(), ptr(batches_tracked),
                                      ctypes.byref(fuse) if fuse is not None else None, ws.data_ptr(), ws.numel(), stream))
        ctx.save_for_backward(x, w, save_mean, save_invstd, mask)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.relu = bool(relu)
        ctx.pdt = pdt
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable  # first-order only: create_graph=True raises instead of returning constants
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, save_mean, save_invstd, mask = ctx.saved_tensors
        C = x.shape[1]
        M = x.numel() // C
        dev = x.device
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        # the gradient in the layout of x (channels-last storage): a no-op when the consumer produced it that way
        if x.dim() == 4:
            dy = dy.contiguous(memory_format=torch.channels_last)
        elif x.dim() == 5:
            dy = dy.contiguous(memory_format=torch.channels_last_3d)
        else:
            dy = dy.contiguous()
        dx = torch.empty_like(x)
        pd = w.dtype if w is not None else torch.float32  # the kernel writes the two [C] gradients in the parameters' dtype
        dgamma = torch.empty(C, dtype=pd, device=dev)
        dbeta = torch.empty(C, dtype=pd, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = BF._workspace(dev, L.btx_bn_workspace_bytes(M, C), stream)
        want_res = ctx.has_res and ctx.needs_input_grad[8]
        dres = None
        fuse = None
        if ctx.relu:
            dres = torch.empty_like(x) if want_res else None  # g = dy where y > 0: the gradient of the residual branch
            fuse = _lib.BnFuse(None, 1, mask.data_ptr(), dres.data_ptr() if dres is not None else None)
        elif want_res:
            dres = dy  # no ReLU behind the add: the residual branch receives dy itself
        _lib.check(L.btx_bn_train_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), _act_code(x.dtype), M, C,
                                      w.data_ptr() if w is not None else None, ctx.pdt, save_mean.data_ptr(),
                                      save_invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                      ctypes.byref(fuse) if fuse is not None else None, ws.data_ptr(), ws.numel(), stream))
        gw = dgamma if (w is not None and ctx.needs_input_grad[1]) else None
        gb = dbeta if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return (dx if ctx.needs_input_grad[0] else None), gw, gb, None, None, None, None, None, dres, None


def batch_norm_train(bn, x, residual=None, relu=False):
    """training-mode forward of the nn.BatchNorm module `bn` through libbtx (bn_train_usable(bn, x) must hold), optionally with the
    block's residual add and ReLU inside the same launches: y = [relu](bn(x) [+ residual])"""
    nbt = bn.num_batches_tracked if bn.track_running_stats else None
    if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64 and nbt.numel() == 1):
        nbt.add_(1)   # not a device int64 word: torch's own increment
        nbt = None
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    # nbt += 1 happens inside the statistics launch
    return BatchNormTrainFn.apply(x, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, nbt, residual, relu)


def bn_act(bn, x, residual=None, relu=True):
    """[relu](bn(x) [+ residual]) as the reference's blocks spell it (resnet_large.py:46-62) — through the fused launches when the
    call qualifies (bn_train_usable, residual of x's shape / dtype / storage layout), else with torch's own ops"""
    if bn_train_usable(bn, x) and (residual is None or (residual.shape == x.shape and residual.dtype == x.dtype and
                                                        residual.stride() == x.stride() and residual.is_cuda)):
        return batch_norm_train(bn, x, residual=residual, relu=relu)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y


def max_pool_train_usable(mp, x):
    """nn.MaxPool2d `mp` on `x` can take btx_maxpool2d_cl_train / _bwd: a CUDA f32 / bf16 tensor in channels-last storage with
    C % 8 == 0 that needs a gradient; square integer window <= 15, dilation 1, floor mode, no indices asked for"""
    if not (isinstance(mp, torch.nn.MaxPool2d) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    if not (torch.is_grad_enabled() and x.requires_grad):
        return False
    sq = lambda v: v if isinstance(v, int) else (v[0] if (len(v) == 2 and v[0] == v[1]) else None)  # noqa: E731
    k, s_, p_, d_ = sq(mp.kernel_size), sq(mp.stride if mp.stride is not None else mp.kernel_size), sq(mp.padding), sq(mp.dilation)
    if None in (k, s_, p_, d_) or d_ != 1 or mp.ceil_mode or mp.return_indices or k > 15 or 2 * p_ > k:
        return False
    n, c, h, w = x.shape
    if c % 8 != 0 or x.numel() == 0 or (h + 2 * p_ - k) // s_ + 1 <= 0 or (w + 2 * p_ - k) // s_ + 1 <= 0:
        return False
    return x.is_contiguous(memory_format=torch.channels_last)


class MaxPool2dTrainFn(torch.autograd.Function):
    """F.max_pool2d on the HIP backend under autograd: one byte per output element (the window position of its maximum) instead of
    ATen's int64 index; forward and backward equal torch's bit for bit (tests/test_gpu_backward.py)"""

    @staticmethod
    def forward(ctx, x, k, s_, p_):
        y, idx = BF.maxpool2d_train_hip(x, k, s_, p_)
        ctx.save_for_backward(idx)
        ctx.geom = (tuple(x.shape), k, s_, p_)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable  # first-order only: create_graph=True raises instead of returning constants
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        shape, k, s_, p_ = ctx.geom
        return BF.maxpool2d_bwd_hip(dy.contiguous(memory_format=torch.channels_last), idx, shape, k, s_, p_), None, None, None


def max_pool_train(mp, x):
    sq = lambda v: v if isinstance(v, int) else v[0]  # noqa: E731
    return MaxPool2dTrainFn.apply(x, sq(mp.kernel_size), sq(mp.stride if mp.stride is not None else mp.kernel_size), sq(mp.padding))


class GraphedTrainStep:
    """forward + loss + backward of `model` on a fixed batch, captured ONCE into a hipGraph and replayed per training step
    (reference README.md:114-125: `output = model(x); kl = get_kl_loss(model); loss = ce(output, y) + kl / batch_size;
    loss.backward()`).  An eager step of a converted ResNet18 dispatches ~590 kernel launches through Python / ATen at ~16 us
    each and is bound by the HOST (profiles/r05_experiments.txt E9); a replay has no host work.  What changes between steps — the MC
    sample index that keys BTX-RNG v1 — lives in one device word that run() rewrites, so every replay draws fresh noise exactly as
    an eager step with set_sample_index(model, s) would (forward, data / weight / rho gradients all read the word).

        step = GraphedTrainStep(model, x, target)            # grads live in p.grad (static tensors, rewritten by every replay)
        for it in range(n): loss = step.run(it); optimizer.step()

    In-place parameter updates between replays are seen (the kernels read mu / rho where they live); x / target are read from the
    tensors given here (copy new batches INTO them).  Layers on padded layouts that need their input gradient (sign tensors keyed
    on the host) cannot be captured and raise.  Drop every reference to the loss / outputs of earlier EAGER steps of this model
    before constructing one (torch: a live autograd graph pins its AccumulateGrad nodes to the stream it ran on)."""

    def __init__(self, model, x, target, loss_fn=None, warmup=2):
        from .models.dnn_to_bnn import get_kl_loss
        if not x.is_cuda:
            raise ValueError("GraphedTrainStep needs CUDA (ROCm) tensors")
        self.model, self.x, self.target = model, x, target
        bs = x.shape[0]
        self.loss_fn = loss_fn or (lambda out, tgt: torch.nn.functional.cross_entropy(out.float(), tgt) + get_kl_loss(model) / bs)
        dev = x.device
        import gc
        gc.collect()  # autograd graphs of earlier eager steps still referenced from garbage would pin AccumulateGrad nodes to the
        torch.cuda.synchronize(dev)  # caller's stream (torch warns that this "may break CUDA graph capture": it does)
        self._layers = [m for m in model.modules() if hasattr(m, "_btx_layer_id")]
        self.sample_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        for m in self._layers:
            m.__dict__["_btx_sample_dev"] = self.sample_dev
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup) + 1):
                self._step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for p_ in model.parameters():
            p_.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = self._step()
        # the graph writes its gradients into these pool tensors on every replay; `optimizer.zero_grad()` (set_to_none=True by
        # default) between replays detaches them from the parameters, so run() attaches them again
        self._grads = [(p_, p_.grad) for p_ in model.parameters() if p_.grad is not None]

    def _step(self):
        for p_ in self.model.parameters():
            p_.grad = None
        _rng.presample(self.model, 0)  # one sampling launch for the forward of every layer (reads the device word)
        out = self.model(self.x)
        if isinstance(out, tuple):
            out = out[0]
        loss = self.loss_fn(out, self.target)
        loss.backward()
        return loss.detach()

    def run(self, sample_idx):
        self.sample_dev.fill_(int(sample_idx) & 0x7FFFFFFF)
        self.graph.replay()
        for p_, g_ in self._grads:
            if p_.grad is not g_:
                p_.grad = g_
        return self.loss

    def close(self):
        for m in self._layers:
            m.__dict__["_btx_sample_dev"] = None
            m.__dict__["_btx_pre"] = None
