import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import bayesian_torch_amd as bt
from bayesian_torch_amd import rng as _rng, mc
if os.environ.get("NO_PRESAMPLE"):
    _rng.presample = lambda *a, **k: None
    _rng._presample = lambda *a, **k: None
dev = torch.device("cuda:0")
r = bench.run_mlp_config(dev)
print(os.environ.get("BTX_NO_DMA"), os.environ.get("NO_PRESAMPLE"), "cfg2 MC-samples/s %.0f  ms %.4f  parity %.2e" % (r["value"], r["ms_per_step"], r["logits_rel_l2_vs_f32_mode"]))
