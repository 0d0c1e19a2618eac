#!/usr/bin/env python3
"""debug: bench model (fused, bf16) with the tap-unrolled kernel vs the run-time-tap kernel, layer by layer"""
import os, sys, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import bayesian_torch_amd as bt
from bayesian_torch_amd import mc

dev = torch.device("cuda:0")
bt.manual_seed(2024)
bt.set_precision("bf16")
fuse = "--no-fuse" not in sys.argv
bs = 64
model = bench.build_model("Flipout", dev, torch.bfloat16, fuse=fuse)
torch.manual_seed(1234)
x = torch.randn(bs, 3, 224, 224).to(dev).to(torch.bfloat16)
layers = [m for m in model.modules() if hasattr(m, "_btx_layer_id")]


def run(env, presample):
    recs = []
    hs = [m.register_forward_hook(lambda mod, i, o, recs=recs: recs.append(o.detach().float().clone())) for m in layers]
    orig = {}
    if fuse:  # forward_fused bypasses forward hooks: wrap
        for m in layers:
            orig[m] = m.forward_fused
            def ff(*a, _m=m, **k):
                o = orig[_m](*a, **k)
                recs.append(o.detach().float().clone())
                return o
            m.forward_fused = ff
    for k, v in env.items():
        os.environ[k] = v
    with torch.no_grad():
        bt.set_sample_index(model, 5, presample=presample)
        out = model(x).float()
    torch.cuda.synchronize()
    for k in env:
        del os.environ[k]
    for h in hs:
        h.remove()
    for m in orig:
        del m.forward_fused
    return out, recs


for presample in (False, True):
    o_old, r_old = run({"BTX_NO_TAPS": "1"}, presample)
    o_new, r_new = run({}, presample)
    print("presample", presample, "logits finite old/new", bool(torch.isfinite(o_old).all()), bool(torch.isfinite(o_new).all()),
          "equal", bool(torch.equal(o_old, o_new)), "layers", len(r_old), len(r_new))
    for i, (a, b) in enumerate(zip(r_old, r_new)):
        if not torch.equal(a, b):
            print("  first mismatch at record", i, tuple(a.shape), "finite old/new", bool(torch.isfinite(a).all()), bool(torch.isfinite(b).all()),
                  "max abs diff", float((a - b).abs().max()), "n bad", int((a != b).sum()))
            bad = (a != b).nonzero()
            print("  first bad idx", bad[0].tolist(), "last", bad[-1].tolist())
            break
# graph mode
kl = 0.0
for lanes in (1, 3):
    for env in ({"BTX_NO_TAPS": "1"}, {}):
        for k, v in env.items():
            os.environ[k] = v
        g = mc.GraphedMC(model, x, kl=kl, lanes=lanes)
        if lanes == 1:
            g.run(5)
        else:
            g.run_many([5, 6, 7])
        torch.cuda.synchronize()
        print("graph lanes", lanes, env, "packed finite", bool(torch.isfinite(g.packed).all()), float(g.packed[:10].sum()))
        g.close()
        for k in env:
            del os.environ[k]
