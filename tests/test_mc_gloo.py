"""N>1 path on CPU: MC samples sharded over a world_size-2 gloo group, one all-reduce; the merged statistics must
equal the single-process run (sample s always uses sample_idx = s, whatever the number of ranks)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class SampleKeyedModel(torch.nn.Module):
    """stands in for a variational model: its logits are a pure function of (x, the pinned MC sample index)"""

    def __init__(self):
        super().__init__()
        self._btx_layer_id = 1
        self._btx_sample = 0
        self.lin = torch.nn.Linear(6, 5)

    def forward(self, x):
        g = torch.Generator().manual_seed(1000 + self._btx_sample)
        self._btx_sample += 1
        return self.lin(x) + torch.randn(x.shape[0], 5, generator=g)


def _worker(rank, world, port, S, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayesian_torch_amd import mc
    torch.manual_seed(0)
    model = SampleKeyedModel()
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(5))
    packed = mc.mc_forward(model, x, S)
    if rank == 0:
        torch.save(packed, out_path)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_mc_sharding_world2_equals_world1(tmp_path):
    sys.path.insert(0, ROOT)
    from bayesian_torch_amd import mc
    S = 5  # odd on purpose: rank 0 takes 3 samples, rank 1 takes 2
    torch.manual_seed(0)
    model = SampleKeyedModel()
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(5))
    single = mc.mc_forward(model, x, S)
    out = str(tmp_path / "packed.pt")
    mp.spawn(_worker, args=(2, _free_port(), S, out), nprocs=2, join=True)
    merged = torch.load(out)
    assert merged.shape == single.shape == (mc.packed_numel(4, 5),)
    assert torch.allclose(merged, single, rtol=1e-6, atol=1e-6)
    u = mc.unpack(merged, 4, 5)
    assert float(u["samples"]) == S
    assert torch.allclose(u["mean_prob"].sum(1), torch.ones(4), atol=1e-5)
    assert (u["mutual_information"] > -1e-6).all()


def test_mc_more_ranks_than_samples(tmp_path):
    """a rank with no sample still joins the collective with zeros"""
    out = str(tmp_path / "packed.pt")
    mp.spawn(_worker, args=(2, _free_port(), 1, out), nprocs=2, join=True)
    merged = torch.load(out)
    assert float(merged[-1]) == 1.0
