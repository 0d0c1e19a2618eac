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


def set_sample_index(model, idx):
    """Pin the Monte-Carlo sample index used by the NEXT forward of every variational layer in `model`."""
    for m in model.modules():
        if hasattr(m, "_btx_layer_id"):
            m._btx_sample = int(idx)
