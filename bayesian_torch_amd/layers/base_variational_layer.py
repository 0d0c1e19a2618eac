"""BaseVariationalLayer_ and the shared implementation of every variational layer.

API mirror of reference `layers/base_variational_layer.py:35-68` (get_kernel_size, dnn_to_bnn_flag, kl_div) plus
`_VariationalNd`, the one implementation behind the 14 public classes (Linear / Conv{1,2,3}d / ConvTranspose{1,2,3}d
x Reparameterization / Flipout).  On a CUDA tensor `forward` is ONE call into libbtx.so (fused sampling +
implicit-GEMM contraction) and `kl_loss` is the HIP KL reduction (recomputed on every call: RNG-free, 8 B/element);
on a CPU tensor it is the ATen op chain of the cited reference method with the reference's torch-generator draw order.
"""
import collections
import warnings
from itertools import repeat

import torch
import torch.nn as nn
from torch.nn import Parameter

from .. import _lib
from .. import functional as BF
from .. import rng as _rng

_BACKEND = "auto"  # "auto": CUDA -> hip (with or without autograd), CPU -> aten ; "hip": CUDA only ; "torch": always aten


def set_backend(name):
    global _BACKEND
    if name not in ("auto", "hip", "torch"):
        raise ValueError("backend must be auto|hip|torch")
    _BACKEND = name


def get_kernel_size(x, n):
    if isinstance(x, collections.abc.Iterable):
        return tuple(x)
    return tuple(repeat(x, n))


class BaseVariationalLayer_(nn.Module):
    def __init__(self):
        super().__init__()
        self._dnn_to_bnn_flag = False

    @property
    def dnn_to_bnn_flag(self):
        return self._dnn_to_bnn_flag

    @dnn_to_bnn_flag.setter
    def dnn_to_bnn_flag(self, value):
        self._dnn_to_bnn_flag = value

    def kl_div(self, mu_q, sigma_q, mu_p, sigma_p):
        """KL(Q||P) of two diagonal Gaussians, MEAN-reduced (reference base_variational_layer.py:53-68)."""
        return BF.kl_aten(mu_q, sigma_q, mu_p, sigma_p)


class _VariationalNd(BaseVariationalLayer_):
    """family: "reparam" | "flipout".  nd: 0 Linear, 1/2/3 Conv.  Draw orders:
    reparam: eps_w, eps_b ; linear flipout: eps_w, eps_b, s_in, s_out ; conv flipout: s_in, s_out, eps_w, eps_b."""

    _family = "reparam"
    _nd = 0
    _transposed = False

    def _setup(self, in_ch, out_ch, kernel_size, stride, padding, dilation, groups, output_padding,
               prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias, check_groups):
        nd = self._nd
        if nd > 0 and check_groups:
            if in_ch % groups != 0:
                raise ValueError('invalid in_channels size')
            if out_ch % groups != 0:
                raise ValueError('invalid in_channels size')
        self.prior_mean = prior_mean
        self.prior_variance = prior_variance
        self.bias = bias
        wn = "weight" if nd == 0 else "kernel"
        self._wn = wn
        if nd == 0:
            self.in_features, self.out_features = in_ch, out_ch
            wshape = (out_ch, in_ch)
            ksz = 1
        else:
            self.in_channels, self.out_channels = in_ch, out_ch
            self.kernel_size = kernel_size
            self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
            if self._transposed:
                self.output_padding = output_padding
            ksz = get_kernel_size(kernel_size, nd)
            if len(ksz) != nd:
                raise ValueError("kernel_size must have %d entries" % nd)
            wshape = ((in_ch, out_ch // groups) if self._transposed else (out_ch, in_ch // groups)) + tuple(ksz)
        self._op = BF.OpDesc(nd, in_ch, out_ch, ksz if nd else 1, stride if nd else 1, padding if nd else 0,
                             dilation if nd else 1, groups if nd else 1, self._transposed,
                             output_padding if (nd and self._transposed) else 0)
        # storage is GEMM-major (what the MFMA kernel streams); the logical shape/values are the reference's
        setattr(self, "mu_" + wn, Parameter(BF.gemm_major_param(wshape, self._op)))
        setattr(self, "rho_" + wn, Parameter(BF.gemm_major_param(wshape, self._op)))
        self.register_buffer("eps_" + wn, torch.Tensor(*wshape), persistent=False)
        self.register_buffer("prior_weight_mu", torch.Tensor(*wshape), persistent=False)
        self.register_buffer("prior_weight_sigma", torch.Tensor(*wshape), persistent=False)
        if bias:
            self.mu_bias = Parameter(torch.Tensor(out_ch))
            self.rho_bias = Parameter(torch.Tensor(out_ch))
            self.register_buffer("eps_bias", torch.Tensor(out_ch), persistent=False)
            self.register_buffer("prior_bias_mu", torch.Tensor(out_ch), persistent=False)
            self.register_buffer("prior_bias_sigma", torch.Tensor(out_ch), persistent=False)
        else:
            self.register_parameter("mu_bias", None)
            self.register_parameter("rho_bias", None)
            self.register_buffer("eps_bias", None, persistent=False)
            self.register_buffer("prior_bias_mu", None, persistent=False)
            self.register_buffer("prior_bias_sigma", None, persistent=False)
        # Channel padding for the MFMA kernels: layers whose C/groups is not a multiple of 8 (the RGB stem, odd Linear
        # sizes) are executed on inputs / parameters zero-padded to the next multiple of 8 so they take the fast
        # granule kernels instead of the element-wise gather kernel.  Every real element still appears exactly once,
        # so the Flipout sign semantics are unchanged; BTX-RNG indices of such a layer refer to the padded layout
        # (materialize_noise maps them back).  groups == 1 only.
        self._btx_cpad = None
        if (groups if nd else 1) == 1 and in_ch % 8 != 0:
            self._btx_cpad = (in_ch + 7) // 8 * 8
            self._op_pad = BF.OpDesc(nd, self._btx_cpad, out_ch, ksz if nd else 1, stride if nd else 1,
                                     padding if nd else 0, dilation if nd else 1, 1, self._transposed,
                                     output_padding if (nd and self._transposed) else 0)
        # MI355X-side state (not part of the reference surface)
        self._btx_layer_id = _rng.next_layer_id()
        self._btx_sample = 0
        self.precision = None  # None -> functional.get_precision(); or "f32" / "bf16"
        self.init_parameters()
        self.quant_prepare = False

    # ---- reference surface --------------------------------------------------------------------------------
    def _w(self):
        return getattr(self, "mu_" + self._wn), getattr(self, "rho_" + self._wn)

    def _mu_rho_init(self):
        mu0, rho0 = self.posterior_mu_init, self.posterior_rho_init
        if isinstance(mu0, tuple):  # the Reparameterization classes keep 1-tuples (reference quirk)
            mu0, rho0 = mu0[0], rho0[0]
        return mu0, rho0

    def init_parameters(self):
        mu, rho = self._w()
        mu0, rho0 = self._mu_rho_init()
        self.prior_weight_mu.fill_(self.prior_mean)
        self.prior_weight_sigma.fill_(self.prior_variance)
        # draw into a contiguous tensor (the reference's generator order), then copy into the strided storage
        mu.data.copy_(torch.empty(mu.shape).normal_(mean=mu0, std=0.1))
        rho.data.copy_(torch.empty(rho.shape).normal_(mean=rho0, std=0.1))
        if self.mu_bias is not None:
            self.prior_bias_mu.fill_(self.prior_mean)
            self.prior_bias_sigma.fill_(self.prior_variance)
            self.mu_bias.data.normal_(mean=mu0, std=0.1)
            self.rho_bias.data.normal_(mean=rho0, std=0.1)

    def _prior_key(self):
        """identity + version of the prior buffers: changes on re-assignment (reference utils/util.py MOPED:
        `layer.prior_weight_mu = det_layer.weight.data`), on `.to(device)` and on in-place writes (`fill_`, `copy_`)"""
        bufs = [self.prior_weight_mu, self.prior_weight_sigma, self.prior_bias_mu, self.prior_bias_sigma]
        return tuple((b.data_ptr(), b._version, tuple(b.shape)) if b is not None else None for b in bufs)

    def _priors_are_scalar(self):
        """True when every prior buffer still holds the constructor's scalars (the HIP KL then reads 8 B/element instead
        of 16).  Verified against the buffers themselves whenever their identity/version changed — never inferred from
        `_version` alone (a moved buffer restarts at 0, an in-place MOPED write can land on the old count).  `.data`
        writes bump nothing: call `layer.refresh_priors()` after one."""
        key = self._prior_key()
        st = self.__dict__.get("_btx_prior_state")
        if st is not None and st[0] == key:
            return st[1]
        ok = bool((self.prior_weight_mu == self.prior_mean).all()) and bool((self.prior_weight_sigma == self.prior_variance).all())
        if ok and self.prior_bias_mu is not None:
            ok = bool((self.prior_bias_mu == self.prior_mean).all()) and bool((self.prior_bias_sigma == self.prior_variance).all())
        self.__dict__["_btx_prior_state"] = (key, ok)
        return ok

    def refresh_priors(self):
        self.__dict__["_btx_prior_state"] = None

    def __deepcopy__(self, memo):
        """copies draw their OWN noise (the reference's copies advance the global generator independently): a deep copy
        gets a fresh BTX-RNG layer id instead of sharing (seed, layer_id, sample) with its source"""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        new.__dict__["_btx_layer_id"] = _rng.next_layer_id()
        for k in ("_btx_pre", "_btx_sample_dev", "_btx_prior_state", "_btx_plans", "_btx_lanes", "_btx_lane_batch"):
            new.__dict__.pop(k, None)
        return new

    def _use_hip(self, t):
        """CUDA (ROCm) tensors run on libbtx — with autograd too (bayesian_torch_amd/autograd.py); CPU tensors and
        backend "torch" take the ATen chain."""
        if _BACKEND == "torch" or not t.is_cuda:
            if _BACKEND == "hip" and not t.is_cuda:
                raise _lib.BtxError("backend 'hip' needs CUDA (ROCm) tensors")
            return False
        return True

    def _needs_grad(self, t):
        mu, rho = self._w()
        return torch.is_grad_enabled() and (mu.requires_grad or rho.requires_grad or
                                            (self.mu_bias is not None and self.mu_bias.requires_grad) or
                                            (t is not None and t.is_floating_point() and t.requires_grad))

    def kl_loss(self):
        mu, rho = self._w()
        if not self._use_hip(mu):
            mu, rho = BF.plain_layout(mu), BF.plain_layout(rho)
            kl = self.kl_div(mu, BF.softplus_naive(rho), self.prior_weight_mu, self.prior_weight_sigma)
            if self.mu_bias is not None:
                kl = kl + self.kl_div(self.mu_bias, BF.softplus_naive(self.rho_bias), self.prior_bias_mu,
                                      self.prior_bias_sigma)
            return kl
        # RNG-free and cheap (8 B/element): recomputed on every call rather than cached — `param.data` mutations do not
        # bump `_version`, so no cache key is trustworthy.  One launch for weight + bias; differentiable when needed.
        from .. import autograd as _ag
        return _ag.kl_of_layers([self])

    def forward(self, input, return_kl=True):
        if self.dnn_to_bnn_flag:
            return_kl = False
        if self._use_hip(input):
            if self._needs_grad(input):
                if self.__dict__.get("_btx_lanes", 1) > 1:
                    raise _lib.BtxError("MC sample lanes are an inference feature: clear them before a training step")
                from .. import autograd as _ag
                s_idx = self._btx_sample
                self.__dict__["_btx_sample"] = s_idx + 1
                mu, rho = self._w()
                out = _ag.ContractFn.apply(self, s_idx, input, mu, rho, self.mu_bias, self.rho_bias)
            else:
                out = self._forward_hip(input)
            if return_kl:
                return out, self.kl_loss()
            return out
        return self._forward_aten(input, return_kl)

    # ---- MI355X path -----------------------------------------------------------------------------------------
    def presample_item(self, sample_idx, prec):
        """What bayesian_torch_amd.presample() needs to sample this layer's weights ahead of its next forward (None
        until the layer has seen an input: the stem layouts depend on the input shape).  Padded layouts (row-fused
        stems, channel padding) are sampled straight from the unpadded parameters (BtxSampleItem.src_KW / src_C)."""
        shape = getattr(self, "_btx_last_xshape", None)
        if shape is None:
            return None
        mu, rho = self._w()
        op0 = self._op
        mu_p, rho_p = BF.gemm_major_view(mu, op0), BF.gemm_major_view(rho, op0)
        plan = None
        if op0.nd == 2 and op0.in_channels <= 4:
            plan = self.__dict__.get("_btx_plans", {}).get(tuple(shape)) or BF.rowfuse_plan(op0, tuple(shape))
        if plan is not None:
            tag, op, src = ("rowfuse", plan["cp"], plan["kwp"]), plan["op"], (plan["kw"], plan["cin"])
        elif self._btx_cpad is not None:
            kw_last = op0.kernel[2] if op0.nd else 1
            tag, op, src = ("cpad", self._btx_cpad), self._op_pad, (kw_last, op0.in_channels)
        else:
            tag, op, src = ("plain",), op0, ()
        if src and op0.transposed:
            return None  # the transposed GEMM-major order is not [N][taps][C] of the padded geometry: per-launch sampling
        # Single-sample launches of Linear layers with at most 256 rows are sampled INSIDE their contraction launch (libbtx routes them to
        # the register-staged kernel: softplus + Philox in registers, no tile in HBM — btx_api.hip "sample where the weights are
        # used"): nothing to pre-sample.  (bf16x3 has no register-staged form and keeps its tiles.)
        if op0.nd == 0 and (self.precision or prec) != "bf16x3" and self._lanes()[0] == 1 and not BF._CONCURRENT:
            rows = 1
            for v in shape[:-1]:
                rows *= int(v)
            if rows <= 256:
                return None
        # Only the LDS-DMA kernel family consumes pre-sampled tiles, and it takes a layer only when a K-stage (32 bf16 /
        # 16 f32 channels) lies inside one filter tap; everything else (depthwise / odd grouped convolutions, C/groups
        # not a multiple of the stage) samples in registers and would ignore — or, for K % 4 != 0, could not even
        # produce — the tiles.
        bk = 32 if (self.precision or prec) == "bf16" else 16
        cg = op.in_channels // op.groups
        if plan is None and cg % bk != 0:
            return None
        kind = _lib.KIND_FLIPOUT if self._family == "flipout" else _lib.KIND_REPARAM
        key = (_rng.seed(), self._sample_key(sample_idx), self._btx_layer_id, self.precision or prec, tag)
        return key, (kind, op, mu_p, rho_p, self._btx_layer_id) + tuple(src)

    def _lanes(self):
        """MC sample lanes of the next forward (rng.set_sample_lanes): (n, images per lane)"""
        return self.__dict__.get("_btx_lanes", 1), self.__dict__.get("_btx_lane_batch")

    def _sample_key(self, sample_idx):
        sdev = getattr(self, "_btx_sample_dev", None)  # graph mode (mc.GraphedMC): the index lives on the device
        n = self.__dict__.get("_btx_lanes", 1)
        return ("dev", sdev.data_ptr(), n) if sdev is not None else (int(sample_idx), n)

    def _take_presampled(self, sample_idx, prec, tag):
        """one-shot: the buffer bayesian_torch_amd.presample() left for exactly this (seed, sample, layer, prec, layout)"""
        pre = self.__dict__.get("_btx_pre")
        self.__dict__["_btx_pre"] = None  # plain attribute: skip nn.Module.__setattr__'s bookkeeping on the hot path
        if pre is not None and pre[0] == (_rng.seed(), self._sample_key(sample_idx), self._btx_layer_id, prec, tag):
            return pre[1]
        return None

    def _forward_hip(self, x, noise=None, sample_idx=None, epilogue=None, gather=False):
        if self._op.nd > 0 and x.dim() == self._op.nd + 1:  # unbatched [C,*sp], as ATen's convolutions accept
            if epilogue is not None and epilogue.get("residual") is not None:
                epilogue = dict(epilogue, residual=epilogue["residual"].unsqueeze(0))
            return self._forward_hip(x.unsqueeze(0), noise, sample_idx, epilogue, gather).squeeze(0)
        mu, rho = self._w()
        mu_p, rho_p = BF.gemm_major_view(mu, self._op), BF.gemm_major_view(rho, self._op)
        lanes, lane_batch = self._lanes()
        if lanes > 1 and noise is not None:
            raise _lib.BtxError("explicit noise is single-sample: clear the sample lanes first (rng.set_sample_lanes)")
        if sample_idx is None:
            sample_idx = self._btx_sample
            self.__dict__["_btx_sample"] = sample_idx + lanes
        kind = _lib.KIND_FLIPOUT if self._family == "flipout" else _lib.KIND_REPARAM
        mb = self.mu_bias.detach() if self.mu_bias is not None else None
        rb = self.rho_bias.detach() if self.rho_bias is not None else None
        op = self._op
        self.__dict__["_btx_last_xshape"] = tuple(x.shape)
        plan = self._rowfuse_plan(x) if noise is None else None
        if plan is not None:  # small-C stem: one kernel row per K-stage on the LDS-DMA kernel
            prec = self.precision or BF.get_precision()
            pre = self._take_presampled(sample_idx, prec, ("rowfuse", plan["cp"], plan["kwp"]))
            # with pre-sampled tiles the launch never reads mu/rho (the row-fused path has no in-register sampler to
            # fall back to), so the padded copies are only made when the launch samples for itself
            mu_f, rho_f = BF.rowfuse_weights(mu_p, rho_p, plan) if pre is None else (mu_p, rho_p)
            # the padded copy is made in the MFMA dtype (the rounding a staging kernel would apply anyway); the
            # output keeps the caller's activation dtype
            xin = self._rowfuse_packed(x, plan, torch.bfloat16 if prec == "bf16" else torch.float32)
            fo = plan["op"].out_spatial((1, plan["Hp"], plan["Wp"]))
            if epilogue is not None and epilogue.get("residual") is not None and fo[2] != plan["Wo"]:
                raise _lib.BtxError("residual epilogue is not available for this row-fused stem geometry")
            out = BF.contract_hip(kind, xin, mu_f, rho_f, mb, rb, plan["op"], _rng.seed(), sample_idx,
                                  self._btx_layer_id, prec=prec, extra_flags=_lib.FLAG_ROWFUSE, out_dtype=x.dtype,
                                  epilogue=epilogue, sampled_w=pre, sample_dev=getattr(self, "_btx_sample_dev", None),
                                  lanes=lanes, lane_batch=lane_batch,
                                  self_sampling_weights=(lambda: BF.rowfuse_weights(mu_p, rho_p, plan)) if pre is not None else None)
            if epilogue is not None and epilogue.get("pool"):
                return out  # pooled inside the launch (pool_fusable() vouched for the geometry)
            return out[:, :, :plan["Ho"], :plan["Wo"]]
        if epilogue is not None and epilogue.get("pool"):
            raise _lib.BtxError("the fused max-pool needs the row-fused stem path (pool_fusable)")
        if self._btx_cpad is not None and noise is None:  # explicit noise (parity mode) stays unpadded -> gather kernel
            extra = self._btx_cpad - op.in_channels
            x = BF.pad_channels(x, op, extra)
            mu_p = torch.nn.functional.pad(mu_p, (0, extra))   # zero weights meet zero activations
            rho_p = torch.nn.functional.pad(rho_p, (0, extra))
            op = self._op_pad
        pre = None
        if noise is None:
            tag = ("cpad", self._btx_cpad) if self._btx_cpad is not None else ("plain",)
            pre = self._take_presampled(sample_idx, self.precision or BF.get_precision(), tag)
        return BF.contract_hip(kind, x, mu_p, rho_p, mb, rb, op, _rng.seed(), sample_idx,
                               self._btx_layer_id, prec=self.precision, noise=noise, epilogue=epilogue, sampled_w=pre,
                               sample_dev=getattr(self, "_btx_sample_dev", None),
                               extra_flags=_lib.FLAG_GATHER if gather else 0, lanes=lanes, lane_batch=lane_batch)

    def _rowfuse_packed(self, x, plan, dtype):
        """the row-fused stem's input in its kernel layout (btx_rowfuse_pack).  mc.GraphedMC(static_input=True) vouches
        that the batch does not change between the MC samples it replays: the pack then happens once, at capture time,
        instead of once per replay (GraphedMC.set_input() re-packs).  The packed copies belong to the GRAPH (`_btx_static_x`
        is that graph's own {layer id: (key, tensor)} store): sibling graphs on one model never share or free each
        other's buffers."""
        store = self.__dict__.get("_btx_static_x")
        if store is None or store is False:
            return BF.rowfuse_input(x, plan, dtype)
        key = (x.data_ptr(), tuple(x.shape), x.dtype, dtype)
        st = store.get(id(self))
        if st is None or st[0] != key:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.BtxError("static_input: the stem input must be packed before the capture starts")
            st = (key, BF.rowfuse_input(x, plan, dtype))
            store[id(self)] = st
        return st[1]

    def _static_repack(self, x, store):
        """mc.GraphedMC.set_input(): refill the packed copy the captured launches read (same buffer, new contents)"""
        st = store.get(id(self))
        if st is None or st[0][0] != x.data_ptr():
            return
        plan = self._rowfuse_plan(x)
        st[1].copy_(BF.rowfuse_input(x, plan, st[1].dtype))

    def pool_fusable(self, x):
        """True when forward_fused(..., pool=True) can fold nn.MaxPool2d(3, 2, 1) into this layer's launch: a row-fused
        bf16 stem on the GPU whose padded geometry has exactly the layer's output extent (btx_contract_pool_shape)."""
        if not self._use_hip(x) or x.dtype != torch.bfloat16 or (self.precision or BF.get_precision()) != "bf16":
            return False
        plan = self._rowfuse_plan(x)
        if plan is None:
            return False
        fo = plan["op"].out_spatial((1, plan["Hp"], plan["Wp"]))
        if (fo[1], fo[2]) != (plan["Ho"], plan["Wo"]):
            return False
        return BF.contract_pool_ok(plan["op"], x.shape[0], (1, plan["Hp"], plan["Wp"]), torch.bfloat16, "bf16",
                                   _lib.FLAG_ROWFUSE)

    def forward_fused(self, x, scale=None, shift=None, residual=None, relu=False, pool=False):
        """`relu?(forward(x) * scale[c] + shift[c] (+ residual))` with the affine / residual / ReLU
        folded into the store of the HIP contraction (eval-mode BatchNorm folds into scale/shift).  Returns `out` only.
        With autograd (any of x / the parameters requires grad) the contraction runs through ContractFn and the affine,
        residual and ReLU as ATen ops, so gradients flow; CPU tensors evaluate the whole expression with ATen ops."""
        if self._use_hip(x) and not self._needs_grad(x):
            return self._forward_hip(x, epilogue=dict(scale=scale, shift=shift, residual=residual, relu=relu, pool=pool))
        if pool:
            raise _lib.BtxError("forward_fused(pool=True) is an inference-only GPU path (pool_fusable)")
        # autograd (a fused model run with grad enabled) or CPU: the same expression as differentiable ops — the
        # contraction through ContractFn on the GPU, the affine / residual / ReLU as ATen ops
        out = self.forward(x, return_kl=False) if self._use_hip(x) else self._forward_aten(x, False)
        shape = (1, -1) + (1,) * self._op.nd if self._op.nd else (1, -1)
        if scale is not None:
            out = out * scale.view(shape).to(out.dtype)
        if shift is not None:
            out = out + shift.view(shape).to(out.dtype)
        if residual is not None:
            out = out + residual
        return torch.relu(out) if relu else out

    def _rowfuse_plan(self, x):
        if self._op.nd != 2 or self._op.in_channels > 4 or not x.is_cuda:
            return None
        cache = self.__dict__.setdefault("_btx_plans", {})  # geometry only: one plan (and one OpDesc) per input shape
        key = tuple(x.shape)
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            cache[key] = BF.rowfuse_plan(self._op, key)
        return cache[key]

    def materialize_noise(self, sample_idx, x_shape=None, out_shape=None, x_dtype=None, signs=True, sample_dev=None):
        """The noise BTX-RNG v1 defines for MC sample `sample_idx` of this layer, in the reference's logical
        layouts: dict(eps_w, eps_b[, sign_in, sign_out]).  Also refreshes the eps_* buffers (the reference's
        observable side effect, conv_variational.py:362).  signs=False: skip the Flipout sign tensors (the backward
        regenerates the hashed signs inside its kernels and only needs eps)."""
        mu, _ = self._w()
        op, seed, lid = self._op, _rng.seed(), self._btx_layer_id
        cin, cpad = op.in_channels, self._btx_cpad
        # sample_dev (captured training steps): the index lives in a device word; eps follows it, the sign TENSORS (padded
        # layouts only) are keyed on the host and cannot
        if sample_dev is not None and signs and self._family == "flipout" and x_shape is not None:
            raise _lib.BtxError("sign tensors of padded layouts need a host-side sample index: this layer cannot run in a captured step")
        _fill_eps = BF.fill_eps_hip
        if sample_dev is not None:
            _fill_eps = lambda n, dev, sd, si, li, st: BF.fill_eps_hip(n, dev, sd, si, li, st, sample_dev=sample_dev)  # noqa: E731
        plan = None
        if x_shape is not None and op.nd == 2 and cin <= 4:
            plan = BF.rowfuse_plan(op, tuple(x_shape))
        if plan is not None:  # indices run over the row-fused layouts: weights [Cout][KH][kwp][cp], input [N][Hp][Wp][cp]
            kh, kw, kwp, cp = op.kernel[1], plan["kw"], plan["kwp"], plan["cp"]
            flat = _fill_eps(mu.shape[0] * kh * kwp * cp, mu.device, seed, sample_idx, lid, _lib.STREAM_EPS_W)
            e = flat.reshape(mu.shape[0], kh, kwp, cp)[:, :, :kw, :cin].permute(0, 3, 1, 2).contiguous()
            d = {"eps_w": e}
            getattr(self, "eps_" + self._wn).copy_(e)
            if self.mu_bias is not None:
                d["eps_b"] = _fill_eps(self.mu_bias.numel(), mu.device, seed, sample_idx, lid, _lib.STREAM_EPS_B)
                self.eps_bias.copy_(d["eps_b"])
            if self._family == "flipout" and signs:
                n, _, h, w = x_shape
                sp = BF.fill_sign_hip(n * plan["Hp"] * plan["Wp"] * cp, mu.device, seed, sample_idx, lid,
                                      _lib.STREAM_SIGN_IN).reshape(n, plan["Hp"], plan["Wp"], cp)
                d["sign_in"] = sp[:, plan["ph"]:plan["ph"] + h, plan["pw"]:plan["pw"] + w, :cin].permute(0, 3, 1, 2)
                # the kernel's output tensor is [N][Ho'][Wo'][Cout] with Wo' possibly wider than Wo
                fo = plan["op"].out_spatial((1, plan["Hp"], plan["Wp"]))
                so = BF.fill_sign_hip(n * fo[1] * fo[2] * op.out_channels, mu.device, seed, sample_idx, lid,
                                      _lib.STREAM_SIGN_OUT).reshape(n, fo[1], fo[2], op.out_channels)
                d["sign_out"] = so[:, :plan["Ho"], :plan["Wo"], :].permute(0, 3, 1, 2)
            return d
        if cpad is None:
            flat = _fill_eps(mu.numel(), mu.device, seed, sample_idx, lid, _lib.STREAM_EPS_W)
        else:  # indices run over the zero-padded [N][tap][cpad] layout
            flat = _fill_eps(mu.numel() // cin * cpad, mu.device, seed, sample_idx, lid, _lib.STREAM_EPS_W)
            flat = flat.reshape(-1, cpad)[:, :cin].reshape(-1)
        d = {"eps_w": BF.unpack_gemm_major(flat, tuple(mu.shape), op)}
        getattr(self, "eps_" + self._wn).copy_(d["eps_w"])
        if self.mu_bias is not None:
            d["eps_b"] = _fill_eps(self.mu_bias.numel(), mu.device, seed, sample_idx, lid, _lib.STREAM_EPS_B)
            self.eps_bias.copy_(d["eps_b"])
        if self._family == "flipout" and x_shape is not None and signs:
            def cl_to_logical(flat8, shape):
                if op.nd == 0:
                    return flat8.reshape(shape)
                n, c, sp = shape[0], shape[1], tuple(shape[2:])
                t = flat8.reshape((n,) + sp + (c,))
                return t.permute((0, op.nd + 1) + tuple(range(1, op.nd + 1)))
            nin = 1
            for v in x_shape:
                nin *= v
            nout = 1
            for v in out_shape:
                nout *= v
            if cpad is None:
                d["sign_in"] = cl_to_logical(BF.fill_sign_hip(nin, mu.device, seed, sample_idx, lid,
                                                              _lib.STREAM_SIGN_IN), tuple(x_shape))
            else:
                sp = BF.fill_sign_hip(nin // cin * cpad, mu.device, seed, sample_idx, lid, _lib.STREAM_SIGN_IN)
                d["sign_in"] = cl_to_logical(sp.reshape(-1, cpad)[:, :cin].reshape(-1), tuple(x_shape))
            d["sign_out"] = cl_to_logical(BF.fill_sign_hip(nout, mu.device, seed, sample_idx, lid,
                                                           _lib.STREAM_SIGN_OUT), tuple(out_shape))
        return d

    # ---- ATen path (CPU tensors / autograd) -------------------------------------------------------------------
    def _forward_aten(self, x, return_kl):
        mu, rho = self._w()
        # the parameters are stored GEMM-major (strided); ATen must see the reference's plain layout or conv picks
        # channels_last outputs and the in-place uniform_() draws land on different logical elements
        mu, rho = BF.plain_layout(mu), BF.plain_layout(rho)
        op = self._op
        eps_w_buf = getattr(self, "eps_" + self._wn)
        has_b = self.mu_bias is not None
        kl = None
        if self._family == "reparam":
            sigma_w = BF.softplus_naive(rho)
            weight = mu + (sigma_w * eps_w_buf.data.normal_())
            if return_kl:
                kl = self.kl_div(mu, sigma_w, self.prior_weight_mu, self.prior_weight_sigma)
            b = None
            if has_b:
                sigma_b = BF.softplus_naive(self.rho_bias)
                b = self.mu_bias + (sigma_b * self.eps_bias.data.normal_())
                if return_kl:
                    kl = kl + self.kl_div(self.mu_bias, sigma_b, self.prior_bias_mu, self.prior_bias_sigma)
            out = BF.contract_aten(x, weight, b, op)
        elif op.nd == 0:  # LinearFlipout: eps first, then the signs
            sigma_w = BF.softplus_naive(rho)
            delta = sigma_w * eps_w_buf.data.normal_()
            if return_kl:
                kl = self.kl_div(mu, sigma_w, self.prior_weight_mu, self.prior_weight_sigma)
            b = None
            if has_b:
                sigma_b = BF.softplus_naive(self.rho_bias)
                b = sigma_b * self.eps_bias.data.normal_()
                if return_kl:
                    kl = kl + self.kl_div(self.mu_bias, sigma_b, self.prior_bias_mu, self.prior_bias_sigma)
            outputs = BF.contract_aten(x, mu, self.mu_bias, op)
            sign_in = x.clone().uniform_(-1, 1).sign()
            sign_out = outputs.clone().uniform_(-1, 1).sign()
            out = outputs + BF.contract_aten(x * sign_in, delta, b, op) * sign_out
        else:  # Conv*Flipout: signs first, then eps
            outputs = BF.contract_aten(x, mu, self.mu_bias, op)
            sign_in = x.clone().uniform_(-1, 1).sign()
            sign_out = outputs.clone().uniform_(-1, 1).sign()
            sigma_w = BF.softplus_naive(rho)
            delta = sigma_w * eps_w_buf.data.normal_()
            if return_kl:
                kl = self.kl_div(mu, sigma_w, self.prior_weight_mu, self.prior_weight_sigma)
            b = None
            if has_b:
                sigma_b = BF.softplus_naive(self.rho_bias)
                b = sigma_b * self.eps_bias.data.normal_()
                if return_kl:
                    kl = kl + self.kl_div(self.mu_bias, sigma_b, self.prior_bias_mu, self.prior_bias_sigma)
            out = outputs + BF.contract_aten(x * sign_in, delta, b, op) * sign_out
        if return_kl:
            return out, kl
        return out


class _VariationalLSTM(BaseVariationalLayer_):
    """LSTM cell unrolled over time on two Bayesian Linear layers (input-to-hidden `ih`, hidden-to-hidden `hh`, both
    in -> 4*out) — reference layers/variational_layers/rnn_variational.py:46-153 and layers/flipout_layers/
    rnn_flipout.py:46-153.  X is [batch, seq, in_features]; every time step calls both Linear layers, i.e. draws fresh
    weight noise per step as the reference does (on a GPU: two fused sample-and-GEMM launches per step, consecutive
    BTX-RNG sample indices).  Returns (hidden_seq, (hidden_seq, cell_seq)[, kl]), kl = the per-step KL terms summed over
    the steps — the reference's accounting.  `_linear_cls` is the Linear class of the family."""

    _linear_cls = None

    def __init__(self, in_features, out_features, prior_mean=0, prior_variance=1, posterior_mu_init=0,
                 posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.prior_mean, self.prior_variance = prior_mean, prior_variance
        self.posterior_mu_init = posterior_mu_init,   # 1-tuples: a quirk of the reference's attribute surface
        self.posterior_rho_init = posterior_rho_init,
        self.bias = bias
        kw = dict(prior_mean=prior_mean, prior_variance=prior_variance, posterior_mu_init=posterior_mu_init,
                  posterior_rho_init=posterior_rho_init, bias=bias)
        self.ih = self._linear_cls(in_features=in_features, out_features=out_features * 4, **kw)
        self.hh = self._linear_cls(in_features=out_features, out_features=out_features * 4, **kw)

    def kl_loss(self):
        return self.ih.kl_loss() + self.hh.kl_loss()

    def forward(self, X, hidden_states=None, return_kl=True):
        if self.dnn_to_bnn_flag:
            return_kl = False
        nb, steps, _ = X.size()
        hs = self.out_features
        for lin in (self.ih, self.hh):
            # every time step must draw fresh noise: eager forwards get it from the layers' advancing sample counter; a
            # device-resident sample index (mc.GraphedMC) or MC sample lanes pin one index for the whole forward, which
            # would reuse one (seed, sample, layer) draw for every step
            if X.is_cuda and steps > 1 and (lin.__dict__.get("_btx_lanes", 1) > 1 or getattr(lin, "_btx_sample_dev", None) is not None):
                raise _lib.BtxError("LSTM layers need a fresh MC sample index per time step, but this layer's index is pinned "
                                    "(MC sample lanes are set, or a live mc.GraphedMC keeps the index in a device word): "
                                    "close() the GraphedMC / rng.set_sample_lanes(model, None), then run eager forwards "
                                    "(mc.mc_forward(lanes=1))")
        if hidden_states is None:
            h_t = torch.zeros(nb, hs, device=X.device, dtype=X.dtype)
            c_t = torch.zeros(nb, hs, device=X.device, dtype=X.dtype)
        else:
            h_t, c_t = hidden_states
        hidden, cells, kl = [], [], 0
        # GPU: the KL terms do not depend on the time step (RNG-free functions of the parameters): one reduction launch per
        # Linear layer per sequence instead of two per step; the per-step accumulation below keeps the reference's
        # summation order (rnn_flipout.py:125-133).  CPU: the reference's own chain, KL inside every Linear forward.
        once = X.is_cuda and self.ih._use_hip(X)
        if once:
            kl_i, kl_h = self.ih.kl_loss(), self.hh.kl_loss()
        for t in range(steps):
            if once:
                gi, gh = self.ih(X[:, t, :], return_kl=False), self.hh(h_t, return_kl=False)
            else:
                gi, kl_i = self.ih(X[:, t, :])
                gh, kl_h = self.hh(h_t)
            gates = gi + gh
            kl = kl + kl_i + kl_h
            i_t, f_t = torch.sigmoid(gates[:, :hs]), torch.sigmoid(gates[:, hs:2 * hs])
            g_t, o_t = torch.tanh(gates[:, 2 * hs:3 * hs]), torch.sigmoid(gates[:, 3 * hs:])
            c_t = f_t * c_t + i_t * g_t
            h_t = o_t * torch.tanh(c_t)
            hidden.append(h_t)
            cells.append(c_t)
        hidden_seq = torch.stack(hidden, dim=1).contiguous()  # [batch, seq, out]
        c_ts = torch.stack(cells, dim=1).contiguous()
        if self._family == "flipout":
            self.kl = kl  # reference rnn_flipout.py:150
        if return_kl:
            return hidden_seq, (hidden_seq, c_ts), kl
        return hidden_seq, (hidden_seq, c_ts)
