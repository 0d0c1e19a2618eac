"""utils.util — the parts of reference utils/util.py on or next to the hot path.

get_rho (MOPED rho init, reference utils/util.py:63-69) is on the path (BASELINE config 5).  The uncertainty
helpers (reference :41-60) are the host-side consumers of the MC outputs; the device-side equivalent that feeds the
RCCL all-reduce is bayesian_torch_amd.mc (btx_mc_accumulate).
"""
import numpy as np
import torch


def get_rho(sigma, delta):
    """rho such that softplus(rho) = delta*|w| : log(expm1(delta*|w|) + 1e-20)."""
    return torch.log(torch.expm1(delta * torch.abs(sigma)) + 1e-20)


def entropy(prob):
    return -1 * np.sum(prob * np.log(prob + 1e-15), axis=-1)


def predictive_entropy(mc_preds):
    """entropy of the MC-mean predictive distribution; mc_preds [S, batch, classes] probabilities."""
    return entropy(np.mean(mc_preds, axis=0))


def mutual_information(mc_preds):
    """BALD: H[mean_s p_s] - mean_s H[p_s]."""
    return entropy(np.mean(mc_preds, axis=0)) - np.mean(entropy(mc_preds), axis=0)
