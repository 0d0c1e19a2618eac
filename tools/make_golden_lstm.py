#!/usr/bin/env python3
"""Generate tests/golden/lstm.npz from the REFERENCE's LSTM layers (layers/variational_layers/rnn_variational.py,
layers/flipout_layers/rnn_flipout.py).  Same recipe as tools/make_golden.py: torch.manual_seed(init) -> reference layer
(its own init draws); x ~ randn; torch.manual_seed(fwd) -> the reference forward (noise from the global generator).  Runs
in the build container only (/root/reference is absent on the GPU box); the fixture is committed.

usage: python tools/make_golden_lstm.py
"""
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")

import bayesian_torch.layers as RL  # noqa: E402  (the reference)

CASES = [("lstm_reparam", "LSTMReparameterization", dict(in_features=12, out_features=10), (4, 6, 12), 11, 22),
         ("lstm_flipout", "LSTMFlipout", dict(in_features=16, out_features=8, bias=False), (3, 5, 16), 33, 44)]


def main():
    out = {}
    for name, cls, kw, xshape, s_init, s_fwd in CASES:
        torch.manual_seed(s_init)
        layer = getattr(RL, cls)(**kw)
        x = torch.randn(*xshape)
        torch.manual_seed(s_fwd)
        with torch.no_grad():
            hs, (hs2, cs), kl = layer(x)
        out[name + "/x"] = x.numpy()
        out[name + "/hidden"] = hs.numpy()
        out[name + "/cells"] = cs.numpy()
        out[name + "/kl"] = np.float32(float(kl))
        out[name + "/kl_loss"] = np.float32(float(layer.kl_loss()))
        for k, v in layer.state_dict().items():
            out[name + "/sd/" + k] = v.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lstm.npz"), **out)
    print("wrote tests/golden/lstm.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
