"""Eval-mode BatchNorm (+ residual, + ReLU) folded into the variational convolutions of a ResNet — SURVEY §8(f)-3,
"the step either side of the path" (reference block structure: models/deterministic/resnet_large.py:46-62, 85-105,
156-171).  `fuse_resnet(model)` rewrites, in place, every block that looks like a torchvision BasicBlock / Bottleneck
(conv1,bn1,conv2,bn2[,conv3,bn3],downsample) whose convs are variational layers, and the stem (conv1,bn1,relu).
The module tree is untouched (state_dict keys of the fused and the unfused model are identical; checkpoints load either
way) and the folded (scale, shift) follow the BatchNorm tensors.  Only for inference (`model.eval()`); the unfused model
is the parity reference (tests/test_gpu_model.py).
"""
import types

import torch
import torch.nn as nn


STEM_POOL_FUSION = True  # conv1 -> bn1 -> relu -> maxpool as one launch when the geometry allows (btx_contract_pool_shape)


def fold_bn(bn):
    """eval-mode BatchNorm as y = x*scale + shift (f32)"""
    with torch.no_grad():
        w = bn.weight.float() if bn.weight is not None else torch.ones_like(bn.running_var, dtype=torch.float32)
        b = bn.bias.float() if bn.bias is not None else torch.zeros_like(bn.running_var, dtype=torch.float32)
        scale = w / torch.sqrt(bn.running_var.float() + bn.eps)
        shift = b - bn.running_mean.float() * scale
    return scale.contiguous(), shift.contiguous()


def _is_var(m):
    return hasattr(m, "forward_fused")


# ---- "is this module's forward the dataflow the fused forms assume?" — decided by RUNNING it, not by its class name -------------
# Attribute names prove nothing about a forward (a pre-activation block has conv1 / bn1 / conv2 / bn2 / downsample too), and a
# class-name list stops at the classes it knows.  A deep copy of the module on the CPU, in eval mode, is run twice on a small random
# input — once through its CLASS's forward, once through the textbook dataflow below — and must agree.  The copy's rho parameters
# are set to -40 first (sigma ~ 4e-18): its variational layers then compute with their means whatever noise they draw, so the two
# runs do not have to call the layers in the same order.  A module that fails the probe is left alone, with a warning.
def _out(o):
    return o[0] if isinstance(o, tuple) else o


def _textbook_block(m, x):
    """reference models/deterministic/resnet_large.py:46-62 (BasicBlock), 85-105 (Bottleneck)"""
    y = torch.relu(m.bn1(_out(m.conv1(x))))
    y = m.bn2(_out(m.conv2(y)))
    if hasattr(m, "conv3") and hasattr(m, "bn3"):
        y = m.bn3(_out(m.conv3(torch.relu(y))))
    idt = x if m.downsample is None else m.downsample(x)
    return torch.relu(y + _out(idt))


def _textbook_resnet(m, x):
    """resnet_large.py:156-171"""
    x = m.maxpool(torch.relu(m.bn1(_out(m.conv1(x)))))
    x = m.layer4(m.layer3(m.layer2(m.layer1(x))))
    return _out(m.fc(m.avgpool(x).flatten(1)))


def _behaves_like(m, ref_fn, shapes):
    import copy
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            probe = copy.deepcopy(m).to("cpu").float().eval()
            with torch.no_grad():
                for name, prm in probe.named_parameters():
                    if name.rsplit(".", 1)[-1].startswith("rho_"):
                        prm.fill_(-40.0)
        except Exception:  # noqa — a module that cannot be copied cannot be probed: not fused
            return False
        for shp in shapes:
            try:
                with torch.no_grad(), torch.random.fork_rng(devices=[]):
                    x = torch.randn(*shp, generator=torch.Generator().manual_seed(7))
                    torch.manual_seed(1234)
                    a = _out(type(probe).forward(probe, x))
                    torch.manual_seed(1234)
                    b = ref_fn(probe, x)
                return bool(a.shape == b.shape and torch.allclose(a, b, rtol=1e-5, atol=1e-6))
            except Exception:  # noqa — e.g. an input too small for the model's pooling: try the next shape
                continue
    return False


def _cin(conv):
    for k in ("in_channels", "in_features"):
        if hasattr(conv, k):
            return int(getattr(conv, k))
    return None


def block_is_textbook(m):
    """conv1 / bn1 / conv2 / bn2 [/ conv3 / bn3] / downsample with the reference's BasicBlock / Bottleneck dataflow (probed, cached)"""
    if "_btx_textbook" in m.__dict__:
        return m.__dict__["_btx_textbook"]
    ok = all(hasattr(m, k) for k in ("conv1", "bn1", "conv2", "bn2", "downsample")) and _cin(m.conv1) is not None
    if ok:
        ok = _behaves_like(m, _textbook_block, [(2, _cin(m.conv1), 8, 8)])
        if not ok:
            import warnings
            warnings.warn("bayesian_torch_amd.models.fuse: %s has conv1/bn1/conv2/bn2/downsample but not the ResNet block dataflow "
                          "(relu(bn(conv)) ... + identity, relu): left unfused" % type(m).__name__)
    object.__setattr__(m, "_btx_textbook", bool(ok))
    return bool(ok)


def resnet_is_textbook(model):
    """conv1 -> bn1 -> relu -> maxpool -> layer1..4 -> avgpool -> flatten -> fc (probed on a 64^2, then a 224^2 input; cached)"""
    if "_btx_textbook" in model.__dict__:
        return model.__dict__["_btx_textbook"]
    ok = all(hasattr(model, k) for k in ("conv1", "bn1", "maxpool", "layer1", "layer2", "layer3", "layer4", "avgpool", "fc"))
    ok = ok and _cin(model.conv1) is not None
    if ok:
        c = _cin(model.conv1)
        ok = _behaves_like(model, _textbook_resnet, [(1, c, 64, 64), (1, c, 224, 224)])
        if not ok:
            import warnings
            warnings.warn("bayesian_torch_amd.models.fuse: %s has a ResNet's attributes but not its forward: stem / head left unfused"
                          % type(model).__name__)
    object.__setattr__(model, "_btx_textbook", bool(ok))
    return bool(ok)


class _Folded:
    """conv (variational) + eval-mode BN folded into the conv's store.  A plain object, NOT an nn.Module: it is attached
    with object.__setattr__, so the module tree — and with it state_dict() / load_state_dict() keys, .to(), .parameters()
    — is exactly that of the unfused model.  (scale, shift) are recomputed whenever the BN tensors change identity or
    version (load_state_dict, .to(device), in-place edits)."""

    def __init__(self, conv, bn):
        self.conv, self.bn = conv, bn
        self._key, self._ss = None, None

    def _scale_shift(self):
        bn = self.bn
        ts = [t for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if key != self._key:
            self._ss, self._key = fold_bn(bn), key
        return self._ss

    def __call__(self, x, residual=None, relu=False, pool=False):
        scale, shift = self._scale_shift()
        return self.conv.forward_fused(x, scale, shift, residual, relu, pool=pool)


def _downsample(self, x):
    f = self.__dict__.get("_fds")
    if f is not None:
        return f(x)
    return x if self.downsample is None else self.downsample(x)


def _basic_forward(self, x):
    idt = _downsample(self, x)
    y = self._f1(x, None, True)
    return self._f2(y, idt, True)


def _bottleneck_forward(self, x):
    idt = _downsample(self, x)
    y = self._f1(x, None, True)
    y = self._f2(y, None, True)
    return self._f3(y, idt, True)


def fuse_resnet(model):
    n = 0
    for m in model.modules():
        names = [k for k in ("conv1", "bn1", "conv2", "bn2") if hasattr(m, k)]
        if (len(names) == 4 and hasattr(m, "downsample") and _is_var(m.conv1) and _is_var(m.conv2)
                and ("_f1" in m.__dict__ or block_is_textbook(m))):
            object.__setattr__(m, "_f1", _Folded(m.conv1, m.bn1))
            object.__setattr__(m, "_f2", _Folded(m.conv2, m.bn2))
            if hasattr(m, "conv3") and _is_var(m.conv3):
                object.__setattr__(m, "_f3", _Folded(m.conv3, m.bn3))
                m.forward = types.MethodType(_bottleneck_forward, m)
            else:
                m.forward = types.MethodType(_basic_forward, m)
            ds = m.downsample  # stays the original nn.Sequential(conv, bn): same keys, the folded call goes beside it
            if isinstance(ds, nn.Sequential) and len(ds) == 2 and _is_var(ds[0]) and isinstance(ds[1], nn.BatchNorm2d):
                object.__setattr__(m, "_fds", _Folded(ds[0], ds[1]))
            n += 1
    # stem: conv1 -> bn1 -> relu -> maxpool
    if (hasattr(model, "conv1") and hasattr(model, "bn1") and hasattr(model, "maxpool") and _is_var(model.conv1)
            and ("_stem" in model.__dict__ or resnet_is_textbook(model))):
        object.__setattr__(model, "_stem", _Folded(model.conv1, model.bn1))

        def pool(mp, y):
            """the stem's max-pool on the channels-last activations (own HBM-bound kernel when the geometry allows)"""
            ints = all(isinstance(v, int) for v in (mp.kernel_size, mp.stride, mp.padding, mp.dilation))
            if (y.is_cuda and isinstance(mp, nn.MaxPool2d) and ints and mp.dilation == 1 and not mp.ceil_mode
                    and not mp.return_indices and y.shape[1] % 8 == 0 and 2 * mp.padding <= mp.kernel_size
                    and y.dtype in (torch.float32, torch.bfloat16) and not torch.is_grad_enabled()):
                from .. import functional as BF
                return BF.maxpool2d_hip(y, mp.kernel_size, mp.stride, mp.padding)
            return mp(y)

        def stem_pool_fusable(self, x):
            """conv1 -> bn1 -> relu -> MaxPool2d(3, 2, 1) in ONE launch (btx_contract_stempool.h): the 112x112 conv
            output never reaches HBM.  Decided per input shape; everything else pools with its own kernel."""
            mp = self.maxpool
            if not STEM_POOL_FUSION:  # A/B measurements (bench.py --no-stem-pool)
                return False
            if not (isinstance(mp, nn.MaxPool2d) and mp.kernel_size in (3, (3, 3)) and mp.stride in (2, (2, 2))
                    and mp.padding in (1, (1, 1)) and mp.dilation in (1, (1, 1)) and not mp.ceil_mode
                    and not mp.return_indices and not torch.is_grad_enabled()):
                return False
            from .. import functional as BF
            cache = self.__dict__.setdefault("_stem_pool_ok", {})
            key = (tuple(x.shape), x.dtype, x.device, self.conv1.precision or BF.get_precision())
            if key not in cache:
                cache[key] = bool(self.conv1.pool_fusable(x))
            return cache[key]

        def fwd(self, x):
            if stem_pool_fusable(self, x):
                x = self._stem(x, None, True, pool=True)
            else:
                x = pool(self.maxpool, self._stem(x, None, True))
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
            ap = self.avgpool
            # the reference's nn.AvgPool2d(7, stride=1) on a 7x7 map (resnet_large.py:125) and torchvision's
            # AdaptiveAvgPool2d((1,1)) are both a global average
            is_global = (isinstance(ap, nn.AdaptiveAvgPool2d) and ap.output_size in (1, (1, 1))) or (
                isinstance(ap, nn.AvgPool2d) and ap.padding in (0, (0, 0)) and not ap.ceil_mode
                and ap.divisor_override is None
                and (ap.kernel_size if isinstance(ap.kernel_size, tuple) else (ap.kernel_size,) * 2) == tuple(x.shape[2:]))
            if (x.is_cuda and is_global and x.shape[1] % 8 == 0
                    and x.dtype in (torch.float32, torch.bfloat16) and not torch.is_grad_enabled()):
                from .. import functional as BF
                return self.fc(BF.avgpool_global_hip(x))
            return self.fc(ap(x).flatten(1))
        model.forward = types.MethodType(fwd, model)
        n += 1
    return n


def _basic_forward_train(self, x):
    if not (self.training and x.is_cuda):
        return self._btx_fwd_eval(x)
    from .. import autograd as _ag
    idt = x if self.downsample is None else self.downsample(x)
    y = _ag.bn_act(self.bn1, self.conv1(x))
    return _ag.bn_act(self.bn2, self.conv2(y), residual=idt)


def _bottleneck_forward_train(self, x):
    if not (self.training and x.is_cuda):
        return self._btx_fwd_eval(x)
    from .. import autograd as _ag
    idt = x if self.downsample is None else self.downsample(x)
    y = _ag.bn_act(self.bn1, self.conv1(x))
    y = _ag.bn_act(self.bn2, self.conv2(y))
    return _ag.bn_act(self.bn3, self.conv3(y), residual=idt)


_BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)


class _Bound:
    """`fn(module, ...)` as the module's instance-level `forward`.  Unlike a closure holding the ORIGINAL module's bound forward,
    copy.deepcopy gives the copy a _Bound on the COPY (an EMA / AveragedModel copy normalising with the original's weights and
    updating the original's running estimates was the bug), and unlike types.MethodType of a module-level function it pickles
    (torch.save(model)): `fn` goes by reference, the module through the pickler's memo."""
    __slots__ = ("fn", "__self__")

    def __init__(self, fn, module):
        self.fn, self.__self__ = fn, module

    def __call__(self, *a, **k):
        return self.fn(self.__self__, *a, **k)

    def __getstate__(self):
        return (self.fn, self.__self__)

    def __setstate__(self, st):
        self.fn, self.__self__ = st


# The fallback of both is the CLASS's forward on this very module.
def _bn_forward(self, x):
    from .. import autograd as _ag
    if _ag.bn_train_usable(self, x):
        return _ag.batch_norm_train(self, x)
    return type(self).forward(self, x)


def _mp_forward(self, x):
    from .. import autograd as _ag
    if _ag.max_pool_train_usable(self, x):
        return _ag.max_pool_train(self, x)
    return type(self).forward(self, x)


def _resnet_forward_train(self, x):
    if not (self.training and x.is_cuda):
        return self._btx_fwd_eval(x)
    from .. import autograd as _ag
    x = self.maxpool(_ag.bn_act(self.bn1, self.conv1(x)))
    x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
    return self.fc(self.avgpool(x).flatten(1))


def _plain_block(m):
    """a block whose forward IS the textbook one — conv / bn pairs, ReLU, `downsample` — decided by running it (block_is_textbook):
    the classes of models/resnet.py and torchvision.models.resnet pass, so does a user's own block with the same dataflow; a block
    that merely has the same attribute names does not"""
    return (isinstance(getattr(m, "relu", None), nn.ReLU)
            and all(isinstance(getattr(m, k, None), nn.modules.batchnorm._BatchNorm) for k in ("bn1", "bn2"))
            and block_is_textbook(m))


def hip_batchnorm(model, fuse_act=True):
    """Route the TRAINING-mode forward (and backward) of every nn.BatchNorm{1,2,3}d of `model` through libbtx
    (csrc/btx_bn.hip) whenever the call qualifies — CUDA, f32 / bf16, channels-last storage, C % 8 == 0
    (autograd.bn_train_usable) — and leave everything else (eval mode, CPU, other layouts) to torch.  The module tree,
    parameters, buffers and state_dict keys are untouched; results match F.batch_norm to rounding.  The reference's training
    loop (README.md:114-125) spends a third of a ResNet18 step in ATen's channels-last BatchNorm kernels.

    fuse_act: the blocks and the stem of models/resnet.py / torchvision.models.resnet (the architecture of the reference's
    models/deterministic/resnet_large.py:46-62, 85-105) also get their `relu(bn(.))` and
    `relu(bn(.) + identity)` inside the normalisation's launches while training on the GPU (one rounding instead of two or three;
    the ReLU's backward mask is a bit per element written by the forward).  Eval mode and CPU tensors keep the forward the block
    had (the eval-mode folding of fuse_resnet included).  nn.MaxPool2d modules are routed the same way where a gradient is
    needed (btx_maxpool2d_cl_train / _bwd).  Returns the number of BatchNorm modules routed."""
    n = 0
    for m in model.modules():
        if type(m) in _BN_TYPES and "_btx_bn_orig" not in m.__dict__:
            # exact nn.BatchNorm{1,2,3}d only: nn.SyncBatchNorm (cross-rank statistics) and the Lazy* variants (parameters
            # that do not exist yet) are _BatchNorm subclasses too and keep torch's own forward
            object.__setattr__(m, "_btx_bn_orig", True)
            m.forward = _Bound(_bn_forward, m)
            n += 1
        elif type(m) is nn.MaxPool2d and "_btx_mp_orig" not in m.__dict__:
            # the pooling layer behind the stem under autograd: btx_maxpool2d_cl_train / _bwd (a byte per output element instead of
            # ATen's int64 indices); everything outside autograd.max_pool_train_usable keeps torch's op
            object.__setattr__(m, "_btx_mp_orig", True)
            m.forward = _Bound(_mp_forward, m)
    if fuse_act:
        for m in model.modules():
            if _plain_block(m) and "_btx_fwd_eval" not in m.__dict__:
                object.__setattr__(m, "_btx_fwd_eval", m.forward)
                three = hasattr(m, "conv3") and hasattr(m, "bn3")
                m.forward = types.MethodType(_bottleneck_forward_train if three else _basic_forward_train, m)
        if (isinstance(getattr(model, "relu", None), nn.ReLU) and "_btx_fwd_eval" not in model.__dict__
                and all(hasattr(model, k) for k in ("conv1", "bn1", "maxpool", "layer1", "layer4", "avgpool", "fc"))
                and resnet_is_textbook(model)):
            object.__setattr__(model, "_btx_fwd_eval", model.forward)
            model.forward = types.MethodType(_resnet_forward_train, model)
    return n
