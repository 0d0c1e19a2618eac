"""Monte-Carlo inference driver: independent MC samples sharded over the GPUs of a node, ONE collective.

What the reference does on one device (examples/main_bayesian_imagenet_dnn2bnn.py:480-499: loop `model(x)`
num_monte_carlo times, torch.stack, softmax, mean; utils/util.py:41-60 for entropy / mutual information) becomes:

  rank r of R runs the samples {s : s mod R == r}; sample s always uses Philox key (seed, sample_idx = s), so the
  result does not depend on R.  Each rank accumulates, on its GPU and in f32, the packed vector
        [ bs*C  sum_s p_s | bs*C  sum_s p_s^2 | bs  sum_s H(p_s) | sum_s KL | number of samples ]
  (btx_mc_accumulate, one launch per sample) and ONE `all_reduce(SUM)` over RCCL/xGMI merges the ranks
  (cfg4: 128 066 floats = 0.5 MB — latency-bound, so one call, not one per tensor).

CPU tensors (gloo tests, config 0) accumulate the same packed layout with ATen ops.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from . import rng as _rng
from .models.dnn_to_bnn import get_kl_loss


def packed_numel(bs, num_classes):
    return 2 * bs * num_classes + bs + 2


MC_MAX_CLASSES = 24575  # btx_mc_accumulate[_lanes] keeps a row of probabilities in LDS (include/btx.h K6)


def _accumulate_torch(packed, logits, kl):
    """the same statistics with torch ops on the tensor's own device: CPU tensors, and GPU rows wider than MC_MAX_CLASSES"""
    bs, C = logits.shape
    p = torch.softmax(logits.float(), dim=1)
    packed[:bs * C] += p.reshape(-1)
    packed[bs * C:2 * bs * C] += (p * p).reshape(-1)
    packed[2 * bs * C:2 * bs * C + bs] += -(p * torch.log(p + 1e-15)).sum(dim=1)
    packed[2 * bs * C + bs] += float(kl)
    packed[2 * bs * C + bs + 1] += 1.0
    return packed


def accumulate(packed, logits, kl=0.0):
    """packed += statistics of one MC sample's logits [bs, C] (in place)."""
    bs, C = logits.shape
    if logits.is_cuda and C <= MC_MAX_CLASSES:
        lg = logits.contiguous()
        if lg.dtype == torch.float32:
            act = _lib.ACT_F32
        elif lg.dtype == torch.bfloat16:
            act = _lib.ACT_BF16
        else:
            lg, act = lg.float(), _lib.ACT_F32
        rc = _lib.lib().btx_mc_accumulate(lg.data_ptr(), bs, C, act, float(kl), packed.data_ptr(),
                                          torch.cuda.current_stream(lg.device).cuda_stream)
        _lib.check(rc)
        return packed
    return _accumulate_torch(packed, logits, kl)


def accumulate_lanes(packed, logits, lanes, kl=0.0):
    """packed += statistics of `lanes` MC samples whose logits sit back to back along the batch axis ([lanes*bs, C]) —
    on the GPU one launch (btx_mc_accumulate_lanes: a workgroup owns a batch row and folds its lanes in order, so the
    result equals `lanes` accumulate() calls bit for bit)."""
    lanes = int(lanes)
    bs = logits.shape[0] // lanes
    if (lanes == 1 or not logits.is_cuda or logits.dtype not in (torch.float32, torch.bfloat16)
            or logits.shape[1] > MC_MAX_CLASSES):
        for k in range(lanes):
            accumulate(packed, logits[k * bs:(k + 1) * bs], kl)
        return packed
    lg = logits.contiguous()
    act = _lib.ACT_F32 if lg.dtype == torch.float32 else _lib.ACT_BF16
    _lib.check(_lib.lib().btx_mc_accumulate_lanes(lg.data_ptr(), lanes, bs, lg.shape[1], act, float(kl), packed.data_ptr(),
                                                  torch.cuda.current_stream(lg.device).cuda_stream))
    return packed


def unpack(packed, bs, num_classes):
    """-> dict(mean_prob [bs,C], var_prob [bs,C], predictive_entropy [bs], mutual_information [bs], kl, samples)."""
    C = num_classes
    n = packed[2 * bs * C + bs + 1]
    mean = (packed[:bs * C] / n).reshape(bs, C)
    ex2 = (packed[bs * C:2 * bs * C] / n).reshape(bs, C)
    mean_h = packed[2 * bs * C:2 * bs * C + bs] / n
    pred_h = -(mean * torch.log(mean + 1e-15)).sum(dim=1)
    return {"mean_prob": mean, "var_prob": (ex2 - mean * mean).clamp_min(0), "predictive_entropy": pred_h,
            "mutual_information": pred_h - mean_h, "kl": packed[2 * bs * C + bs] / n, "samples": n}


@torch.no_grad()
def mc_forward(model, x, num_samples, sample_offset=0, with_kl=False, group=None, reduce=True, lanes=1, rank=None,
               world=None):
    """Run `num_samples` MC forward passes of `model` on `x`, sharded over the ranks of `group` (or the default
    group when torch.distributed is initialised), and return the all-reduced packed statistics tensor.
    lanes > 1 (GPU): this rank evaluates its samples `lanes` at a time, one launch per layer for all of them
    (rng.set_sample_lanes) — the same numbers as one at a time.  rank / world: override the process group's (tests that
    play several ranks in one process, with reduce=False)."""
    r0, w0 = 0, 1
    if dist.is_available() and dist.is_initialized():
        r0, w0 = dist.get_rank(group), dist.get_world_size(group)
    rank = r0 if rank is None else int(rank)
    world = w0 if world is None else int(world)
    packed = None
    kl = float(get_kl_loss(model)) if with_kl else 0.0  # RNG-free: identical for every sample
    mine = list(range(rank, num_samples, world))
    lanes = max(1, int(lanes)) if x.is_cuda else 1
    bs = x.shape[0]
    from . import functional as BF
    for g0 in range(0, len(mine), lanes):
        grp = [sample_offset + s for s in mine[g0:g0 + lanes]]
        if len(grp) > 1:
            _rng.set_sample_lanes(model, grp, batch=bs, presample=True)
            logits = model(x)
        elif x.is_cuda:
            if lanes > 1:
                _rng.set_sample_lanes(model, None)
            # a ragged last group of ONE sample is planned like a lane (BTX_FLAG_CONCURRENT: the K split of the throughput
            # plan), so a sample's f32 summation order does not depend on how the samples were grouped over launches / ranks
            with BF.concurrent_plan(lanes > 1 or BF._CONCURRENT):
                _rng.set_sample_index(model, grp[0], presample=True)
                logits = model(x)
        else:
            # CPU tensors take the ATen route, whose noise comes from torch's generator (the reference's draw order): key
            # that generator on (seed, sample index) for the duration of the forward, so that sample s is the same draw on
            # whichever rank evaluates it — the property the GPU path has by construction (BTX-RNG v1); the caller's CPU
            # generator state is restored by fork_rng and no CUDA generator is touched (torch.manual_seed would reseed them)
            _rng.set_sample_index(model, grp[0])
            with torch.random.fork_rng(devices=[]):
                torch.default_generator.manual_seed(_rng.cpu_sample_seed(grp[0]))  # the CPU generator ONLY
                logits = model(x)
        if isinstance(logits, tuple):
            logits = logits[0]
        if packed is None:
            packed = torch.zeros(packed_numel(bs, logits.shape[1]), dtype=torch.float32, device=logits.device)
        accumulate_lanes(packed, logits, len(grp), kl)
    if lanes > 1:
        _rng.set_sample_lanes(model, None)
    if packed is None:  # this rank got no sample: it still takes part in the collective
        with torch.no_grad():
            _rng.set_sample_index(model, sample_offset)
            shape = model(x).shape
        packed = torch.zeros(packed_numel(*shape), dtype=torch.float32, device=x.device)
    if reduce and dist.is_available() and dist.is_initialized() and (w0 > 1 or world == w0):
        # also in a 1-rank group: the packed vector takes the same route (RCCL on a GPU) whatever the rank count
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed


@torch.no_grad()
def mc_forward_batched(model, x, num_samples, chunk=4, sample_offset=0, with_kl=False):
    """Opt-in batched-MC Flipout: the reference's own trick `data = torch.cat([data]*S, 0)` then ONE
    forward (examples/main_bayesian_flipout_imagenet.py:613-618).  `chunk` replicas of the batch go through the model
    together: they share one weight perturbation (one sampling pass, `chunk` x larger pixel tiles) and are decorrelated
    by their per-example Flipout signs only — pseudo-independent samples, NOT the statistics of `mc_forward`, which
    draws fresh weights per sample.  Returns the packed statistics of all `num_samples` replicas (this rank only)."""
    bs = x.shape[0]
    packed = None
    kl = float(get_kl_loss(model)) if with_kl else 0.0
    done, cid = 0, 0
    while done < num_samples:
        c = min(chunk, num_samples - done)
        xb = torch.cat([x] * c, 0) if c > 1 else x
        _rng.set_sample_index(model, sample_offset + cid, presample=xb.is_cuda)
        logits = model(xb)
        if isinstance(logits, tuple):
            logits = logits[0]
        if packed is None:
            packed = torch.zeros(packed_numel(bs, logits.shape[1]), dtype=torch.float32, device=logits.device)
        for i in range(c):
            accumulate(packed, logits[i * bs:(i + 1) * bs], kl)
        done += c
        cid += 1
    return packed


class GraphedMC:
    """One Monte-Carlo sample — weight sampling, the model forward, the accumulation of the predictive statistics —
    captured ONCE into a hipGraph (torch.cuda.CUDAGraph) and replayed per sample.

    A converted ResNet18 step is ~50 kernel launches of 20-70 us; issued one by one from Python the host cannot keep
    the GPU busy.  The graph removes the per-launch host work; what changes between samples — the sample index that
    keys BTX-RNG v1 — lives in one device word (BtxRng.sample_idx_dev) that `run()` rewrites before each replay, so
    every replay draws the noise of ITS sample index exactly as an eager forward with set_sample_index() would.

        g = GraphedMC(model, x, kl=float(get_kl_loss(model)))
        for s in my_samples: g.run(s)
        stats = unpack(g.packed, *g.logits_shape)

    The parameters must not be re-allocated while the graph is alive.  In-place parameter updates are seen by the replays
    of a one-sample graph and of lane_mode "streams"; lane_mode "launch" (the default for lanes > 1) caches the Flipout mean
    tiles and sigma between replays: call refresh_weights() after an update.  The input is read from `self.x` on every
    replay unless static_input=True (then call set_input() to change it)."""

    def __init__(self, model, x, kl=0.0, warmup=2, lanes=1, keep_logits=False, concurrent_hint=None, lane_mode="launch",
                 static_input=False, capture_stream=None):
        """lanes > 1: one replay evaluates `lanes` MC samples (independent noise: the same results as one at a time); use
        run_many().  lane_mode "launch" (default): the samples are lanes of ONE launch per layer (rng.set_sample_lanes:
        4x the workgroups per launch fill the 256 CUs where one sample of a 7x7 / 14x14 layer cannot, and one
        workgroup's prologue / store overlaps another's MFMA loop); the mean tiles of the Flipout layers are written
        once (refresh_weights() after a parameter update), a replay samples sigma*eps only.  "streams": each sample on
        its own stream inside the graph, one launch per (layer, sample) — the round-2 form, kept for A/B.
        static_input (lane_mode "launch"): the input batch is the same for every replay, so a row-fused stem packs it into
        its kernel layout ONCE, outside the graph, instead of once per replay (set_input() re-packs)."""
        if not x.is_cuda:
            raise ValueError("GraphedMC needs CUDA (ROCm) tensors")
        if lane_mode not in ("launch", "streams"):
            raise ValueError("lane_mode must be 'launch' or 'streams'")
        self.model, self.x, self.kl, self.lanes = model, x, float(kl), int(lanes)
        self.lane_mode = lane_mode
        self._capture_stream = capture_stream  # graphs that will be replayed concurrently must not share a capture stream:
        # the contraction workspace (split-K partials) is per stream and its address is baked into the captured launches
        self.static_input = bool(static_input) and self.lanes > 1 and lane_mode == "launch"
        self._static_packs = {}  # this graph's packed stem inputs {id(layer): (key, tensor)}: owned here, freed by close()
        if self.lanes > 1 and lane_mode == "launch":
            self._init_launch_lanes(warmup, keep_logits)
            return
        # keep_logits: lane_logits[k] is the (static) logits tensor of lane k — after a replay it holds the logits of the
        # sample that lane just evaluated (parity tests of the graphed configuration)
        self.keep_logits, self.lane_logits = bool(keep_logits), [None] * int(lanes)
        dev = x.device
        self.sample_devs = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(self.lanes)]
        self.sample_dev = self.sample_devs[0]
        self._layers = [m for m in model.modules() if hasattr(m, "_btx_layer_id")]
        self.packed = None
        self._lane_packed = [None] * self.lanes
        self._streams = [torch.cuda.Stream(dev) for _ in range(self.lanes)] if self.lanes > 1 else []
        from . import functional as BF
        # several samples in flight: plan every launch for device throughput (no split-K through HBM to fill idle CUs)
        with BF.concurrent_plan((self.lanes > 1) if concurrent_hint is None else bool(concurrent_hint)):
            self._capture_streams(dev, warmup)
        for pk in self._lane_packed:
            pk.zero_()
        for m in self._layers:
            m._btx_sample_dev = self.sample_dev

    def _capture_streams(self, dev, warmup):
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup) + 1):  # 1st pass records the input shapes presample() needs
                for k in range(self.lanes):
                    self._lane(k)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. the RCCL watchdog of torch.distributed) keep making runtime calls while this
        # thread captures
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            if self.lanes == 1:
                self._lane(0)
            else:
                main = torch.cuda.current_stream(dev)
                for k, st in enumerate(self._streams):
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        self._lane(k)
                for st in self._streams:
                    main.wait_stream(st)
                for k in range(1, self.lanes):  # fold the lanes' statistics into lane 0's buffer
                    self._lane_packed[0].add_(self._lane_packed[k])
                    self._lane_packed[k].zero_()

    # ---- lane_mode "launch": the MC samples of a replay are lanes of every layer's launch -----------------------------
    def _init_launch_lanes(self, warmup, keep_logits):
        model, x, dev = self.model, self.x, self.x.device
        self.keep_logits, self.lane_logits = bool(keep_logits), [None] * self.lanes
        self._layers = [m for m in model.modules() if hasattr(m, "_btx_layer_id")]
        self.sample_dev = torch.zeros(self.lanes, dtype=torch.int32, device=dev)
        self.sample_devs = [self.sample_dev]
        self._tiles = {}  # tile buffers of this graph (rng._presample cache): the mean tiles live here between replays
        self.packed, self._streams = None, []
        self.bs = x.shape[0]
        self._set_lane_state(True)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for i in range(max(1, warmup) + 1):  # 1st pass records the input shapes presample() needs
                self._launch_lanes(skip_mu=False)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        kw = dict(stream=self._capture_stream) if self._capture_stream is not None else {}
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local", **kw):
            self._launch_lanes(skip_mu=True)
        self.packed.zero_()
        # the replays need nothing from the Python-side lane state: a plain `model(x)` between replays is a plain forward
        self._set_lane_state(False)

    def _set_lane_state(self, on):
        """(re)apply / clear this graph's lane state on the layers WITHOUT touching the sample words: what makes
        rng.presample() address this graph's tile buffers (its cache key holds the lane count and the layers' ids)"""
        for m in self._layers:
            d = m.__dict__
            if on:
                d["_btx_lanes"], d["_btx_lane_batch"], d["_btx_sample_dev"] = self.lanes, self.bs, self.sample_dev
                d["_btx_static_x"] = self._static_packs if self.static_input else None
            else:
                d.pop("_btx_lanes", None)
                d.pop("_btx_lane_batch", None)
                d.pop("_btx_static_x", None)
                d["_btx_sample_dev"] = None
            d["_btx_pre"] = None

    def _launch_lanes(self, skip_mu):
        _rng.presample(self.model, 0, cache=self._tiles, skip_mu=skip_mu)
        logits = self.model(self.x)
        if isinstance(logits, tuple):
            logits = logits[0]
        bs = self.bs
        if self.packed is None:
            self.logits_shape = (bs, logits.shape[1])
            self.packed = torch.zeros(packed_numel(bs, logits.shape[1]), dtype=torch.float32, device=logits.device)
        accumulate_lanes(self.packed, logits, self.lanes, self.kl)  # one launch for all lanes of the replay
        if self.keep_logits:
            for k in range(self.lanes):
                self.lane_logits[k] = logits[k * bs:(k + 1) * bs]

    def refresh_weights(self):
        """lane_mode "launch": rewrite the cached mean tiles and sigma after an in-place parameter update (the replays
        sample sigma*eps only).  The sample words keep their values.  Works whatever happened to the model since the
        capture (plain forwards, a training step, another GraphedMC): the lane state of THIS graph is re-applied for the
        call, so the sampling pass writes the buffers the captured launches read."""
        if self.lanes > 1 and self.lane_mode == "launch":
            if not self._tiles:
                raise _lib.BtxError("GraphedMC.refresh_weights() after close()")
            with torch.no_grad():
                self._set_lane_state(True)
                try:
                    n_before = len(self._tiles)
                    _rng.presample(self.model, 0, cache=self._tiles, skip_mu=False)
                    if len(self._tiles) != n_before:  # a cache miss would fill NEW buffers the replays never read
                        raise _lib.BtxError("GraphedMC.refresh_weights(): the model's layers no longer match the captured graph")
                finally:
                    self._set_lane_state(False)

    def set_input(self, x):
        """copy a new batch (same shape / dtype) into the captured input; with static_input the row-fused stem's packed
        copy — what the captured launches read — is refilled from it"""
        self.x.copy_(x)
        if self.static_input:
            with torch.no_grad():
                for m in self._layers:
                    if hasattr(m, "_static_repack"):
                        m._static_repack(self.x, self._static_packs)

    def _lane(self, k):
        for m in self._layers:
            m.__dict__["_btx_sample_dev"] = self.sample_devs[k]
        _rng.presample(self.model, 0)
        logits = self.model(self.x)
        if isinstance(logits, tuple):
            logits = logits[0]
        if self._lane_packed[k] is None:
            self.logits_shape = tuple(logits.shape)
            self._lane_packed[k] = torch.zeros(packed_numel(*logits.shape), dtype=torch.float32, device=logits.device)
            if k == 0:
                self.packed = self._lane_packed[0]
        accumulate(self._lane_packed[k], logits, self.kl)
        if self.keep_logits:
            self.lane_logits[k] = logits

    def run(self, sample_idx):
        if self.lanes != 1:
            raise ValueError("this graph evaluates %d samples per replay: use run_many()" % self.lanes)
        self.sample_dev.fill_(int(sample_idx) & 0x7FFFFFFF)
        self.graph.replay()

    def run_many(self, sample_indices):
        """one replay = len(sample_indices) == lanes MC samples"""
        if len(sample_indices) != self.lanes:
            raise ValueError("expected %d sample indices" % self.lanes)
        if self.lane_mode == "launch" and self.lanes > 1:
            self.sample_dev.copy_(torch.tensor([int(i) & 0x7FFFFFFF for i in sample_indices], dtype=torch.int32))
        else:
            for w, i in zip(self.sample_devs, sample_indices):
                w.fill_(int(i) & 0x7FFFFFFF)
        self.graph.replay()

    def close(self):
        for m in self._layers:
            m._btx_sample_dev = None
            m._btx_pre = None
            m.__dict__.pop("_btx_lanes", None)
            m.__dict__.pop("_btx_lane_batch", None)
            if m.__dict__.get("_btx_static_x") is self._static_packs:
                m.__dict__.pop("_btx_static_x", None)
        self._static_packs.clear()  # only this graph's buffers: sibling graphs keep theirs
        self._tiles = {}
