"""GPU: SURVEY §8(e) as far as ONE GPU allows — the packed statistics of mc_forward travel through an RCCL all-reduce
(torch.distributed backend "nccl" = RCCL on ROCm) in a 1-rank process group, on the device, and come back unchanged;
bench.py launched by torch.distributed.run with one rank initialises RCCL and puts its all-reduce inside the timed region
(the N > 1 path with N = 1).  Reference analogue being replaced: examples/main_bayesian_imagenet_dnn2bnn.py:156
(DataParallel) and the MC loop :480-499."""
import json
import os
import socket
import subprocess
import sys
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIOR = dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, moped_delta=0.5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_mc_forward_through_a_one_rank_rccl_group():
    import torch.distributed as dist
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import mc
    from bayesian_torch_amd.models import resnet
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        bt.manual_seed(2024)
        bt.set_precision("bf16")
        torch.manual_seed(0)
        m = resnet.resnet18()
        bt.dnn_to_bnn(m, dict(PRIOR, type="Flipout", moped_enable=False))
        m = m.to(dev).eval()
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.to(torch.bfloat16)
        bt.assign_layer_ids(m)
        torch.manual_seed(1234)
        x = torch.randn(8, 3, 224, 224, device=dev).to(torch.bfloat16)
        S = 6
        calls = []
        orig = dist.all_reduce

        def spy(t, *a, **k):
            calls.append((t.device.type, t.dtype, t.numel()))
            return orig(t, *a, **k)
        dist.all_reduce = spy
        try:
            red = mc.mc_forward(m, x, S, with_kl=True, reduce=True, lanes=3)
        finally:
            dist.all_reduce = orig
        loc = mc.mc_forward(m, x, S, with_kl=True, reduce=False, lanes=3)
        torch.cuda.synchronize()
        # exactly ONE collective, on the device, over the whole packed vector
        assert calls == [("cuda", torch.float32, mc.packed_numel(8, 1000))], calls
        assert torch.equal(red, loc), float((red - loc).abs().max())
        u = mc.unpack(red, 8, 1000)
        assert abs(float(u["samples"]) - S) < 0.5 and torch.isfinite(u["mean_prob"]).all()
        assert abs(float(u["kl"]) - float(bt.get_kl_loss(m))) < 1e-3
        # the graphed configuration's buffer through the same collective
        g = mc.GraphedMC(m, x, kl=float(bt.get_kl_loss(m)), lanes=3)
        g.run_many([0, 1, 2])
        g.run_many([3, 4, 5])
        before = g.packed.clone()
        dist.all_reduce(g.packed, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert torch.equal(before, g.packed)
        assert torch.allclose(g.packed, red, rtol=1e-5, atol=1e-6)  # same samples, same lanes => the same statistics
        g.close()
    finally:
        bt.set_precision("f32")
        dist.destroy_process_group()


def test_bench_one_rank_under_torchrun_uses_rccl():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (how the driver launches N > 1, here N = 1):
    the run initialises RCCL, keeps the one all-reduce inside the timed region and reports rccl_ranks == 1; its rate equals
    the plain `python bench.py --gpus 1` invocation within noise."""
    common = ["--gpus", "1", "--steps", "8", "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--no-traffic",
              "--no-launch-timing", "--no-sustain"]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + common
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d.get("collective") == "rccl all_reduce in the timed region"
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, cwd=ROOT, env=env, capture_output=True,
                        text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    d2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    assert d2.get("collective") in (None, "none (1 process, no group)")
    print("bench 8 samples: torchrun 1 rank + RCCL %.1f MC-samples/s, plain %.1f" % (d["value"], d2["value"]))
    assert 0.7 < d["value"] / d2["value"] < 1.3
