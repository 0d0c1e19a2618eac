"""GPU: MC sample lanes (btx_contract_fwd_lanes / btx_sample_weights_lanes, ABI 5) and the tall-strip tiles of the
tap-unrolled kernel.

A lane must compute BIT FOR BIT what a single-sample launch with its sample index computes: the noise indices are
relative to the lane's own tensors, the accumulation order of an output element does not depend on the tiling.  That is
what lets MC samples be evaluated 4 at a time per launch, and sharded over ranks, without the results depending on how
they were grouped."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


LAYER_CASES = [
    # (class, kwargs, input shape per lane, what it exercises)
    ("Conv2dFlipout", dict(in_channels=64, out_channels=64, kernel_size=3, padding=1, bias=False), (4, 64, 56, 56)),   # taps, tall 9x28 strips
    ("Conv2dFlipout", dict(in_channels=64, out_channels=128, kernel_size=3, padding=1, bias=True), (3, 64, 28, 28)),   # taps, tall 9 rows
    ("Conv2dFlipout", dict(in_channels=128, out_channels=64, kernel_size=3, padding=1, bias=False), (5, 128, 14, 14)),  # tall 18 rows, 4 channel blocks
    ("Conv2dFlipout", dict(in_channels=64, out_channels=64, kernel_size=3, padding=1, bias=False), (40, 64, 7, 7)),    # tall 36 rows over many images
    ("Conv2dFlipout", dict(in_channels=32, out_channels=64, kernel_size=3, padding=1, bias=False), (2, 32, 13, 11)),   # odd extents
    ("Conv2dReparameterization", dict(in_channels=64, out_channels=64, kernel_size=3, padding=1, bias=True), (4, 64, 28, 28)),
    # wide Reparameterization tile (64 px x 128 ch per wave): the 3-lane launch has enough workgroups to take it, a single-sample
    # launch of the same layer keeps the narrow tile — same K order, so still bit for bit
    ("Conv2dReparameterization", dict(in_channels=64, out_channels=128, kernel_size=3, padding=1, bias=True), (64, 64, 28, 28)),
    ("Conv2dReparameterization", dict(in_channels=128, out_channels=256, kernel_size=3, padding=1, bias=False), (48, 128, 28, 28)),
    # the same for the stride-2 kernel (contract_taps2_kernel<..., WIDE>): 256 pixel tiles per lane, 3 lanes -> wide; alone -> narrow
    ("Conv2dReparameterization", dict(in_channels=64, out_channels=128, kernel_size=3, stride=2, padding=1, bias=True), (64, 64, 56, 56)),
    ("Conv2dFlipout", dict(in_channels=64, out_channels=128, kernel_size=3, stride=2, padding=1, bias=False), (4, 64, 28, 28)),  # taps2
    ("Conv2dFlipout", dict(in_channels=64, out_channels=128, kernel_size=1, stride=2, bias=False), (4, 64, 28, 28)),   # LDS-DMA kernel
    ("Conv2dFlipout", dict(in_channels=32, out_channels=32, kernel_size=5, padding=2, bias=False), (2, 32, 12, 12)),    # run-time-tap patch kernel
    ("Conv2dFlipout", dict(in_channels=3, out_channels=64, kernel_size=7, stride=2, padding=3, bias=False), (2, 3, 64, 64)),  # row-fused stem
    # row-fused stem whose per-lane output is NOT a 16-byte multiple (1 x 9 x 9 x 6 bf16 = 972 B): the lanes fall back to
    # single-sample launches, which must sample from the PADDED parameters although pre-sampled tiles were handed over
    ("Conv2dFlipout", dict(in_channels=3, out_channels=6, kernel_size=7, stride=2, padding=3, bias=False), (1, 3, 18, 18)),
    ("Conv2dFlipout", dict(in_channels=64, out_channels=256, kernel_size=1, bias=False), (4, 64, 28, 28)),              # pointwise GEMM, resident stages
    ("Conv2dFlipout", dict(in_channels=256, out_channels=128, kernel_size=1, bias=True), (3, 256, 14, 14)),            # pointwise GEMM, streamed stages
    ("Conv2dFlipout", dict(in_channels=128, out_channels=256, kernel_size=1, stride=2, bias=False), (4, 128, 28, 28)),  # 8-wave GEMM, strided
    ("LinearFlipout", dict(in_features=512, out_features=1000), (8, 512)),
    ("LinearReparameterization", dict(in_features=128, out_features=64), (8, 128)),
    ("Conv2dFlipout", dict(in_channels=20, out_channels=24, kernel_size=3, padding=1), (2, 20, 9, 9)),                 # channel-padded -> gather/regstage
]


@pytest.mark.parametrize("prec,act", [("bf16", torch.bfloat16), ("f32", torch.float32), ("bf16x3", torch.float32)])
@pytest.mark.parametrize("case", LAYER_CASES, ids=[c[0] + str(c[2]) for c in LAYER_CASES])
def test_lanes_equal_single_sample_launches(case, prec, act):
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import layers as L
    cls, kw, xshape = case
    dev = _dev()
    bt.manual_seed(1234)
    torch.manual_seed(7)
    layer = getattr(L, cls)(**kw).to(dev)
    layer.precision = prec
    S = 3
    bs = xshape[0]
    shared = cls.endswith("Flipout") and kw.get("in_channels") in (3, 20)  # first-layer style: one input for all lanes
    xs = [torch.randn(*xshape, device=dev).to(act) for _ in range(1 if shared else S)]
    if len(xshape) == 4:
        xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    idx = [11, 12, 40]
    with torch.no_grad():
        singles = []
        from bayesian_torch_amd import functional as BF
        with BF.concurrent_plan():  # the K split of a launch with lanes is that of the throughput plan of ONE lane
            for l in range(S):
                bt.set_sample_lanes(layer, None)
                singles.append(layer._forward_hip(xs[0 if shared else l], sample_idx=idx[l]).float())
        for presample in (False, True):
            x_all = xs[0] if shared else torch.cat(xs, 0)
            if len(xshape) == 4:
                x_all = x_all.contiguous(memory_format=torch.channels_last)
            bt.set_sample_lanes(layer, idx, batch=bs)
            if presample:
                layer._forward_hip(x_all)   # records the input shape presample_item needs
                bt.set_sample_lanes(layer, idx, batch=bs, presample=True)
            out = layer._forward_hip(x_all).float()
            assert out.shape[0] == S * bs
            for l in range(S):
                assert torch.equal(out[l * bs:(l + 1) * bs], singles[l]), "lane %d differs (presample=%s)" % (l, presample)
        bt.set_sample_lanes(layer, None)


@pytest.mark.parametrize("cin,cout,hw,bs", [(64, 64, 56, 8), (128, 128, 28, 8), (256, 256, 14, 16), (512, 512, 7, 64), (64, 64, 30, 3)])
def test_tall_strip_tiles_vs_oracle_chain(cin, cout, hw, bs):
    """the tall-strip tiling (tile rows / widths that do not divide the map) against the reference op chain evaluated by
    torch in f32 with the noise BTX-RNG v1 defines: same bars as tests/test_gpu_at_size.py"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import layers as L
    from oracle import bt_ref
    dev = _dev()
    bt.manual_seed(5)
    torch.manual_seed(3)
    layer = L.Conv2dFlipout(cin, cout, 3, padding=1, bias=True).to(dev)
    x = torch.randn(bs, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    for prec, bar in (("f32", 1e-5), ("bf16", 1e-2)):
        layer.precision = prec
        xin = x if prec == "f32" else x.to(torch.bfloat16)
        with torch.no_grad():
            out = layer._forward_hip(xin, sample_idx=9).float()
            nz = layer.materialize_noise(9, tuple(x.shape), tuple(out.shape), x.dtype)
            mu, rho = layer._w()
            ref = bt_ref.flipout_forward(xin.float(), mu, rho, layer.mu_bias, layer.rho_bias, nz["eps_w"], nz.get("eps_b"),
                                         nz["sign_in"].float(), nz["sign_out"].float(), dict(kind="conv", nd=2, stride=(1, 1), padding=(1, 1),
                                                                             dilation=(1, 1), groups=1))
        rel = float((out - ref).norm() / ref.norm())
        assert rel < bar, (prec, rel)


def test_graphed_mc_launch_lanes_equal_eager_samples():
    """mc.GraphedMC(lanes=4, lane_mode="launch") on the fused ResNet18: the per-sample logits of a replay are those of
    eager single-sample forwards (bit-exact), with and without cached mean tiles"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc
    from bayesian_torch_amd.models.resnet import resnet18
    from bayesian_torch_amd.models.fuse import fuse_resnet
    dev = _dev()
    bt.manual_seed(99)
    torch.manual_seed(0)
    m = resnet18()
    bt.dnn_to_bnn(m, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type="Flipout",
                          moped_enable=False, moped_delta=0.5))
    m = m.to(dev).eval()
    for mod in m.modules():  # bf16 activations: the stock BN layers follow, the variational parameters stay f32
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.to(torch.bfloat16)
    bt.assign_layer_ids(m)
    bt.set_precision("bf16")
    fuse_resnet(m)
    x = torch.randn(8, 3, 224, 224, device=dev).to(torch.bfloat16)
    idx = [3, 4, 21, 22]
    try:
        from bayesian_torch_amd import functional as BF
        with torch.no_grad(), BF.concurrent_plan():
            eager = []
            for i in idx:
                bt.set_sample_index(m, i, presample=True)
                eager.append(m(x).float().clone())
        g = mc.GraphedMC(m, x, kl=0.0, lanes=4, keep_logits=True)
        for rep in range(2):  # second replay: the mean tiles are the cached ones
            g.run_many(idx)
            torch.cuda.synchronize()
            for l in range(4):
                assert torch.equal(g.lane_logits[l].float(), eager[l]), (rep, l)
        stats = mc.unpack(g.packed, 8, 1000)
        assert abs(float(stats["samples"]) - 8.0) < 0.5
        g.close()
        # and the eager lanes driver
        packed = mc.mc_forward(m, x, 6, sample_offset=100, lanes=4, reduce=False)
        with BF.concurrent_plan():
            ref = mc.mc_forward(m, x, 6, sample_offset=100, lanes=1, reduce=False)
        assert torch.allclose(packed, ref, rtol=1e-6, atol=1e-7)
    finally:
        bt.set_precision("f32")


def test_rank_partition_of_samples_is_rank_count_independent():
    """SURVEY §8e on the real HIP ResNet18: R emulated ranks each run {s : s mod R == r} (mc_forward(reduce=False, rank,
    world)); the SUM of their packed vectors — what the one RCCL all-reduce computes — equals the single-rank result.
    Per-sample logits are bit-identical whatever R (previous test); the sums differ by f32 summation order only."""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc
    from bayesian_torch_amd.models.resnet import resnet18
    from bayesian_torch_amd.models.fuse import fuse_resnet
    dev = _dev()
    bt.manual_seed(5)
    torch.manual_seed(1)
    m = resnet18()
    bt.dnn_to_bnn(m, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type="Flipout",
                          moped_enable=False, moped_delta=0.5))
    m = m.to(dev).eval()
    for mod in m.modules():  # bf16 activations: the stock BN layers follow, the variational parameters stay f32
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.to(torch.bfloat16)
    bt.assign_layer_ids(m)
    bt.set_precision("bf16")
    fuse_resnet(m)
    x = torch.randn(4, 3, 224, 224, device=dev).to(torch.bfloat16)
    S = 16
    try:
        one = mc.mc_forward(m, x, S, with_kl=True, reduce=False, lanes=4, rank=0, world=1)
        for R in (2, 4, 8):
            tot = torch.zeros_like(one)
            for r in range(R):
                tot += mc.mc_forward(m, x, S, with_kl=True, reduce=False, lanes=4 if R < 8 else 2, rank=r, world=R)
            assert abs(float(tot[-1]) - S) < 0.5
            assert torch.allclose(tot, one, rtol=1e-5, atol=1e-6), R
    finally:
        bt.set_precision("f32")


def _small_net(dev):
    import bayesian_torch_amd as bt
    torch.manual_seed(4)
    net = torch.nn.Sequential(torch.nn.Conv2d(32, 64, 3, padding=1, bias=False), torch.nn.ReLU(),
                              torch.nn.Conv2d(64, 64, 3, padding=1, bias=True), torch.nn.ReLU(),
                              torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(64, 16))
    bt.dnn_to_bnn(net, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type="Flipout",
                            moped_enable=False, moped_delta=0.5))
    net = net.to(dev).eval()
    bt.assign_layer_ids(net)
    return net


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_graphed_mc_refresh_weights_after_parameter_update(prec):
    """lane_mode "launch" caches the Flipout mean tiles and sigma inside the graph.  After an in-place parameter
    update, refresh_weights() must make the replays see the new parameters whatever ran on the model in between (plain
    forwards, another graph), and the capture must not leave the lane state on the model."""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc, functional as BF
    dev = _dev()
    bt.manual_seed(8)
    bt.set_precision(prec)
    try:
        net = _small_net(dev)
        x = torch.randn(4, 32, 12, 12, device=dev)
        idx = [5, 6, 7]

        def eager():
            outs = []
            with torch.no_grad(), BF.concurrent_plan():
                for i in idx:
                    bt.set_sample_index(net, i, presample=True)
                    outs.append(net(x).float().clone())
            return outs
        g = mc.GraphedMC(net, x, kl=0.0, lanes=3, keep_logits=True)
        with torch.no_grad():
            assert net(x).shape[0] == 4  # a plain forward between replays is a plain forward (no lanes left on the model)
        before = eager()
        g.run_many(idx)
        torch.cuda.synchronize()
        for l in range(3):
            assert torch.equal(g.lane_logits[l].float(), before[l])
        with torch.no_grad():
            for p_ in net.parameters():
                p_.add_(0.05 * torch.randn_like(p_))
        g2 = mc.GraphedMC(net, x, kl=0.0, lanes=2)  # another graph on the same model: its own tile buffers
        after = eager()
        assert not torch.equal(after[0], before[0])
        g.run_many(idx)  # stale tiles: still the old mean weights (documented: call refresh_weights())
        torch.cuda.synchronize()
        assert not torch.equal(g.lane_logits[0].float(), after[0])
        g.refresh_weights()
        g.run_many(idx)
        torch.cuda.synchronize()
        for l in range(3):
            assert torch.equal(g.lane_logits[l].float(), after[l]), l
        g.close()
        g2.close()
        with pytest.raises(Exception):
            g.refresh_weights()
    finally:
        bt.set_precision("f32")


def test_lanes_with_unaligned_lane_strides_fall_back_to_single_launches():
    """a 10-class head at an odd batch has per-lane outputs that are not multiples of 16 bytes —
    btx_contract_fwd_lanes refuses them (BTX_E_ALIGN); the host runs such a layer lane by lane instead of raising"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import layers as L, functional as BF
    dev = _dev()
    bt.manual_seed(3)
    torch.manual_seed(1)
    layer = L.LinearFlipout(64, 10).to(dev)
    bs, idx = 3, [2, 9, 10]
    xs = [torch.randn(bs, 64, device=dev) for _ in idx]
    with torch.no_grad():
        with BF.concurrent_plan():
            singles = [layer._forward_hip(xs[l], sample_idx=idx[l]) for l in range(3)]
        bt.set_sample_lanes(layer, idx, batch=bs)
        out = layer._forward_hip(torch.cat(xs, 0))
        bt.set_sample_lanes(layer, None)
    assert out.shape == (9, 10)
    for l in range(3):
        assert torch.equal(out[l * bs:(l + 1) * bs], singles[l])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_mc_accumulate_lanes_equals_sequential_accumulates(dt):
    from bayesian_torch_amd import mc
    dev = _dev()
    torch.manual_seed(0)
    lanes, bs, C = 5, 7, 1000
    lg = (3 * torch.randn(lanes * bs, C, device=dev)).to(dt)
    a = torch.zeros(mc.packed_numel(bs, C), device=dev)
    b = torch.zeros_like(a)
    for rep in range(2):
        mc.accumulate_lanes(a, lg, lanes, 1.5)
        for k in range(lanes):
            mc.accumulate(b, lg[k * bs:(k + 1) * bs], 1.5)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert float(a[-1]) == 2 * lanes


def test_lstm_refuses_device_resident_sample_index():
    """every LSTM time step must draw fresh noise; a pinned device-resident sample index cannot provide that"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import layers as L
    from bayesian_torch_amd._lib import BtxError
    dev = _dev()
    torch.manual_seed(0)
    lstm = L.LSTMFlipout(16, 16).to(dev)
    X = torch.randn(2, 3, 16, device=dev)
    with torch.no_grad():
        lstm(X)
        sdev = bt.set_sample_lanes(lstm, [1, 2], batch=2)
        with pytest.raises(BtxError):
            lstm(X)
        bt.set_sample_lanes(lstm, None)
        lstm(X)


@pytest.mark.parametrize("prec,act", [("bf16", torch.bfloat16), ("f32", torch.float32)])
def test_mlp_lanes_share_the_input_of_the_first_linear_layer(prec, act):
    """BASELINE cfg2 as bench.py runs it: the MC samples of a LinearFlipout MLP as lanes of one launch per layer.  The FIRST layer
    reads one input for all lanes (a Linear layer with a shared input: its output stacks the lanes along the leading axis — this used
    to fold back into the input's shape and raise); every lane's logits equal a single-sample forward of the throughput plan bit for
    bit, and the graphed replay's statistics equal the sum over those forwards."""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc
    from bayesian_torch_amd import functional as BF
    dev = _dev()
    bt.manual_seed(2024)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(784, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(),
                              torch.nn.Linear(512, 10))
    bt.dnn_to_bnn(net, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type="Flipout",
                            moped_enable=False, moped_delta=0.5))
    net = net.to(dev).eval()
    bt.assign_layer_ids(net)
    bt.set_precision(prec)
    try:
        x = torch.randn(256, 784, device=dev).to(act)
        idx = [3, 9, 10, 77]
        with torch.no_grad():
            singles = []
            with BF.concurrent_plan():
                for s in idx:
                    bt.set_sample_index(net, s)
                    singles.append(net(x).float().clone())
            bt.set_sample_lanes(net, idx, batch=256)
            y = net(x).float()
            bt.set_sample_lanes(net, None)
            assert y.shape == (4 * 256, 10)
            for l in range(4):
                assert torch.equal(y[l * 256:(l + 1) * 256], singles[l]), l
            want = torch.zeros(mc.packed_numel(256, 10), dtype=torch.float32, device=dev)
            for t in singles:
                mc.accumulate(want, t.to(act), 0.25)
            g = mc.GraphedMC(net, x, kl=0.25, lanes=4)
            g.run_many(idx)
            torch.cuda.synchronize()
            assert torch.allclose(g.packed, want, rtol=1e-6, atol=1e-6)
            g.close()
    finally:
        bt.set_precision("f32")
