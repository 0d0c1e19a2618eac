"""Host side of BTX-RNG v1: which (seed, sample_idx, layer_id) a forward call uses.

The reference draws eps / signs from torch's global generator (conv_variational.py:362, conv_flipout.py:385-392):
every forward advances one process-wide stream.  Here the noise is a pure function of
(seed, sample_idx, layer_id, element index), so the only host state is
  * a process-wide seed            (`manual_seed`; defaults to torch.initial_seed() at first use)
  * a per-layer id                 (assigned at construction, re-assignable with `assign_layer_ids(model)`)
  * a per-layer forward counter    (the Monte-Carlo sample index; auto-increments, or pinned with
                                    `set_sample_index(model, s)` — that is how MC samples are sharded over ranks
                                    with results independent of the number of ranks).
"""
import itertools

import torch

_seed = None
_layer_counter = itertools.count(1)


def manual_seed(seed):
    global _seed
    _seed = int(seed) & 0xFFFFFFFFFFFFFFFF


def seed():
    global _seed
    if _seed is None:
        _seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
    return _seed


def next_layer_id():
    return next(_layer_counter)


def assign_layer_ids(model, start=1):
    """Deterministically renumber the variational layers of `model` in module order (call it on every rank)."""
    i = start
    for m in model.modules():
        if hasattr(m, "_btx_layer_id"):
            m._btx_layer_id = i
            i += 1
    return i


def set_sample_index(model, idx, presample=False):
    """Pin the Monte-Carlo sample index used by the NEXT forward of every variational layer in `model`.
    presample=True additionally samples the weights of all layers for that index in ONE launch (see presample())."""
    devs = {}
    for m in model.modules():
        if hasattr(m, "_btx_layer_id"):
            m._btx_sample = int(idx)
            sdev = getattr(m, "_btx_sample_dev", None)  # a live mc.GraphedMC keeps the index on the device: keep it in step
            if sdev is not None:
                devs[sdev.data_ptr()] = sdev
    for sdev in devs.values():
        sdev.fill_(int(idx) & 0x7FFFFFFF)
    if presample:
        _presample(model, idx)


def _presample(model, idx):
    """Sample the weights of every variational layer of `model` for MC sample `idx` in one kernel launch
    (btx_sample_weights) instead of one small pre-pass per layer.  The buffers are consumed by the next forward of each
    layer if — and only if — it runs with the same (seed, sample index, layer id, precision, weight layout); call it
    after the last parameter update before that forward.  Layers that have not seen an input yet, CPU layers and
    explicit-noise calls simply sample in their own launch."""
    from . import functional as BF
    groups = {}
    for m in model.modules():
        if not hasattr(m, "presample_item") or not hasattr(m, "_btx_layer_id"):
            continue
        m._btx_pre = None
        if not next(m.parameters()).is_cuda:
            continue
        it = m.presample_item(idx, BF.get_precision())
        if it is None:
            continue
        key, item = it
        groups.setdefault((key[3], next(m.parameters()).device), []).append((m, key, item))
    for (prec, device), lst in groups.items():
        bufs = BF.sample_weights([it for _, _, it in lst], seed(), idx, prec, device,
                                 sample_dev=getattr(lst[0][0], "_btx_sample_dev", None))
        for (m, key, _), buf in zip(lst, bufs):
            m._btx_pre = (key, buf)


presample = _presample
