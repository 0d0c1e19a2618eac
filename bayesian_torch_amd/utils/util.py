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


def MOPED(model, det_model, det_checkpoint, delta):
    """Empirical-Bayes priors and posterior init from a trained deterministic model (reference utils/util.py:72-136,
    Krishnan et al., AAAI 2020): for every variational layer of `model`, paired by position with the layer of
    `det_model`, prior mean <- w_det (a full-shape TENSOR prior: the KL kernel then reads it, btx_kl_gauss
    prior_mu_t), mu <- w_det, rho <- get_rho(w_det, delta); BatchNorm parameters and statistics are copied.
    `det_checkpoint`: path for torch.load, a state_dict, or None (det_model already holds its weights)."""
    if det_checkpoint is not None:
        sd = torch.load(det_checkpoint) if isinstance(det_checkpoint, (str, bytes)) or hasattr(det_checkpoint, "read") \
            else det_checkpoint
        det_model.load_state_dict(sd)
    for layer, det_layer in zip(model.modules(), det_model.modules()):
        name = layer.__class__.__name__
        if name.endswith("Reparameterization") or name.endswith("Flipout"):
            wn = getattr(layer, "_wn", "weight" if name.startswith("Linear") else "kernel")
            w = det_layer.weight.data
            # in-place copies keep the storage layout the kernels stream (GEMM-major) and the buffer identities
            layer.prior_weight_mu.copy_(w)
            getattr(layer, "mu_" + wn).data.copy_(w)
            getattr(layer, "rho_" + wn).data.copy_(get_rho(w, delta))
            if layer.mu_bias is not None and getattr(det_layer, "bias", None) is not None:
                b = det_layer.bias.data
                layer.prior_bias_mu.copy_(b)
                layer.mu_bias.data.copy_(b)
                layer.rho_bias.data.copy_(get_rho(b, delta))
            if hasattr(layer, "refresh_priors"):
                layer.refresh_priors()
        elif name.startswith("Batch"):
            # through the tensors themselves (not .data): the in-place writes bump _version, which is what the folded
            # BatchNorm of models.fuse keys its cached (scale, shift) on
            with torch.no_grad():
                layer.weight.copy_(det_layer.weight)
                if layer.bias is not None:
                    layer.bias.copy_(det_layer.bias)
                layer.running_mean.copy_(det_layer.running_mean)
                layer.running_var.copy_(det_layer.running_var)
                layer.num_batches_tracked.copy_(det_layer.num_batches_tracked)
    return model
