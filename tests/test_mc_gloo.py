"""N>1 path on CPU: MC samples sharded over a world_size-2 gloo group, one all-reduce; the merged statistics must
equal the single-process run (sample s always uses sample_idx = s, whatever the number of ranks)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


PRIOR = dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, moped_enable=False,
             moped_delta=0.5)


def _real_model(typ="Flipout"):
    """a converted net of REAL variational layers (Conv2dFlipout / LinearFlipout, or the Reparameterization classes) on the
    ATen route — the reference's own op chain and torch-generator draw order; mc_forward keys that generator on the MC
    sample index, which is what makes the sharded result independent of the rank count on CPU as BTX-RNG v1 does on the GPU"""
    import bayesian_torch_amd as bt
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, stride=2),
                              torch.nn.ReLU(), torch.nn.Flatten(), torch.nn.Linear(8 * 3 * 3, 5))
    bt.dnn_to_bnn(net, dict(PRIOR, type=typ))
    net.eval()
    bt.assign_layer_ids(net)
    return net


def _input():
    return torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(5))


def _worker(rank, world, port, S, out_path, typ="Flipout"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc
    bt.manual_seed(2024)
    model = _real_model(typ)
    torch.manual_seed(100 + rank)  # the ranks' own generator states differ: the samples must not depend on them
    packed = mc.mc_forward(model, _input(), S, with_kl=True)
    if rank == 0:
        torch.save(packed, out_path)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single(S, typ="Flipout"):
    sys.path.insert(0, ROOT)
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc
    bt.manual_seed(2024)
    model = _real_model(typ)
    torch.manual_seed(7)
    state = torch.get_rng_state()
    packed = mc.mc_forward(model, _input(), S, with_kl=True)
    assert torch.equal(torch.get_rng_state(), state)  # the caller's generator is left alone
    return packed


def test_mc_sharding_world2_equals_world1(tmp_path):
    """real Conv2dFlipout / LinearFlipout layers, 5 MC samples (rank 0 takes 3, rank 1 takes 2) + the summed KL: the one
    all-reduce of the two ranks' packed vectors equals the single-process result to 1e-6"""
    from bayesian_torch_amd import mc
    S = 5
    for typ in ("Flipout", "Reparameterization"):
        single = _single(S, typ)
        out = str(tmp_path / ("packed_%s.pt" % typ))
        mp.spawn(_worker, args=(2, _free_port(), S, out, typ), nprocs=2, join=True)
        merged = torch.load(out)
        assert merged.shape == single.shape == (mc.packed_numel(4, 5),)
        assert torch.allclose(merged, single, rtol=1e-6, atol=1e-6), (typ, float((merged - single).abs().max()))
        u = mc.unpack(merged, 4, 5)
        assert float(u["samples"]) == S
        assert torch.allclose(u["mean_prob"].sum(1), torch.ones(4), atol=1e-5)
        assert (u["mutual_information"] > -1e-6).all()
        assert float(u["kl"]) > 0  # the summed KL travels in the same vector
        # the samples really are stochastic: two different sample sets give different statistics
        assert not torch.allclose(_single(S, typ)[:20], mc.mc_forward(_real_model(typ), _input(), S, sample_offset=50)[:20])


def test_mc_more_ranks_than_samples(tmp_path):
    """a rank with no sample still joins the collective with zeros"""
    out = str(tmp_path / "packed.pt")
    mp.spawn(_worker, args=(2, _free_port(), 1, out), nprocs=2, join=True)
    merged = torch.load(out)
    assert float(merged[-1]) == 1.0
