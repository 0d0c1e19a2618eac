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


def cpu_sample_seed(sample_idx):
    """torch-generator seed of MC sample `sample_idx` on the ATen (CPU) route: mc.mc_forward forks the generator and seeds it
    with this for the duration of one forward, so a sample is the same draw on whichever rank evaluates it"""
    x = (seed() ^ (0x9E3779B97F4A7C15 * (int(sample_idx) + 1))) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 31
    return x & 0x7FFFFFFFFFFFFFFF


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


def set_sample_lanes(model, indices, batch=None, presample=False, sample_dev=None):
    """MC sample lanes: the NEXT forward of every variational layer of `model` evaluates len(indices) Monte-Carlo samples
    in one launch per layer (btx_contract_fwd_lanes).  Feed the model its input ONCE (`batch` images: the first layers
    read it for every lane) — every activation behind the first variational layer, and the model's output, hold the
    lanes back to back along the batch axis ([len(indices) * batch, ...]).  Lane l computes bit for bit what a plain
    forward with set_sample_index(model, indices[l]) planned for throughput (functional.concurrent_plan(), i.e.
    BTX_FLAG_CONCURRENT — a launch with lanes always takes that plan's K split) would; against the default latency plan
    the f32 summation order of small-map layers can differ by rounding.  `indices=None` (or one index) returns to plain
    forwards.
    GPU-only; the indices live in a device tensor (`sample_dev`: an int32 tensor to (re)use — mc.GraphedMC keeps one
    per graph and rewrites it between replays)."""
    import torch
    layers = [m for m in model.modules() if hasattr(m, "_btx_layer_id")]
    if indices is None or len(indices) <= 1:
        for m in layers:
            m.__dict__.pop("_btx_lanes", None)
            m.__dict__.pop("_btx_lane_batch", None)
            if sample_dev is None:
                m.__dict__["_btx_sample_dev"] = None
        if indices:
            set_sample_index(model, indices[0], presample=presample)
        return None
    if batch is None:
        raise ValueError("set_sample_lanes needs the number of images per lane (batch)")
    n = len(indices)
    dev = next(layers[0].parameters()).device
    if sample_dev is None:
        sample_dev = torch.zeros(n, dtype=torch.int32, device=dev)
    if sample_dev.numel() != n or sample_dev.dtype != torch.int32:
        raise ValueError("sample_dev must be an int32 tensor with one word per lane")
    sample_dev.copy_(torch.tensor([int(i) & 0x7FFFFFFF for i in indices], dtype=torch.int32), non_blocking=True)
    for m in layers:
        m.__dict__["_btx_lanes"] = n
        m.__dict__["_btx_lane_batch"] = int(batch)
        m.__dict__["_btx_sample_dev"] = sample_dev
        m._btx_sample = int(indices[0])
    if presample:
        _presample(model, int(indices[0]))
    return sample_dev


def _presample(model, idx, cache=None, skip_mu=False):
    """Sample the weights of every variational layer of `model` for MC sample `idx` in one kernel launch
    (btx_sample_weights; with sample lanes set: for every lane) instead of one small pre-pass per layer.  skip_mu (needs
    `cache`): the mean tiles of the Flipout layers are left as the previous call with this cache wrote them — the caller
    vouches that mu has not changed since (an MC loop over frozen parameters).  The buffers are consumed by the next forward of each
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
        lanes = lst[0][0].__dict__.get("_btx_lanes", 1)
        # `cache` (a dict the caller owns, e.g. one per mc.GraphedMC): the tile buffers are reused from call to call —
        # what skip_mu needs: the mean tiles an earlier call wrote are still there.  Without it every call gets fresh
        # buffers from torch's allocator (safe across streams).
        old = None
        if cache is not None:
            ckey = (prec, str(device), lanes, tuple((id(m), key[4]) for m, key, _ in lst))
            old = cache.get(ckey)
        bufs = BF.sample_weights([it for _, _, it in lst], seed(), idx, prec, device,
                                 sample_dev=getattr(lst[0][0], "_btx_sample_dev", None), lanes=lanes,
                                 bufs=old, skip_mu=bool(skip_mu and old is not None))
        if cache is not None and old is None:
            cache[ckey] = bufs
        for (m, key, _), buf in zip(lst, bufs):
            m._btx_pre = (key, buf)


presample = _presample
