"""dnn_to_bnn() / get_kl_loss() — host-side API surface of the hot path (reference models/dnn_to_bnn.py:52-165).

Behaviour kept from the reference: recursion into any child that has children; a leaf is converted when its CLASS
NAME contains "Conv" or "Linear" (so ConvTranspose* and look-alike user classes match too) and the new class is
looked up as `<ClassName><params["type"]>` in `bayesian_torch_amd.layers`; the required dict keys are
prior_mu, prior_sigma, posterior_mu_init, posterior_rho_init, type, moped_enable (KeyError if absent), moped_delta;
`output_padding` / `padding_mode` are NOT forwarded; converted layers get `dnn_to_bnn_flag = True` (forward returns
only `out`).  nn.LSTM children become LSTM<type> (two Bayesian Linear layers per time step).
"""
import torch

from .. import layers as bayesian_layers
from ..utils.util import get_rho


def _moped(bnn_layer, d, params, wname):
    delta = params["moped_delta"]
    getattr(bnn_layer, "mu_" + wname).data.copy_(d.weight.data)
    getattr(bnn_layer, "rho_" + wname).data.copy_(get_rho(d.weight.data, delta))
    if bnn_layer.mu_bias is not None:
        bnn_layer.mu_bias.data.copy_(d.bias.data)
        bnn_layer.rho_bias.data.copy_(get_rho(d.bias.data, delta))


def bnn_linear_layer(params, d):
    layer_fn = getattr(bayesian_layers, d.__class__.__name__ + params["type"])
    bnn_layer = layer_fn(in_features=d.in_features, out_features=d.out_features,
                         prior_mean=params["prior_mu"], prior_variance=params["prior_sigma"],
                         posterior_mu_init=params["posterior_mu_init"],
                         posterior_rho_init=params["posterior_rho_init"], bias=d.bias is not None)
    if params["moped_enable"]:
        _moped(bnn_layer, d, params, "weight")
    bnn_layer.dnn_to_bnn_flag = True
    return bnn_layer


def bnn_conv_layer(params, d):
    layer_fn = getattr(bayesian_layers, d.__class__.__name__ + params["type"])
    bnn_layer = layer_fn(in_channels=d.in_channels, out_channels=d.out_channels, kernel_size=d.kernel_size,
                         stride=d.stride, padding=d.padding, dilation=d.dilation, groups=d.groups,
                         prior_mean=params["prior_mu"], prior_variance=params["prior_sigma"],
                         posterior_mu_init=params["posterior_mu_init"],
                         posterior_rho_init=params["posterior_rho_init"], bias=d.bias is not None)
    if params["moped_enable"]:
        _moped(bnn_layer, d, params, "kernel")
    bnn_layer.dnn_to_bnn_flag = True
    return bnn_layer


def bnn_lstm_layer(params, d):
    """reference models/dnn_to_bnn.py:106-122: nn.LSTM(input_size, hidden_size) -> LSTM<type>(in, out); MOPED is not
    defined for LSTM layers (the reference prints the same warning and converts with the default init)."""
    layer_fn = getattr(bayesian_layers, d.__class__.__name__ + params["type"])
    bnn_layer = layer_fn(in_features=d.input_size, out_features=d.hidden_size, prior_mean=params["prior_mu"],
                         prior_variance=params["prior_sigma"], posterior_mu_init=params["posterior_mu_init"],
                         posterior_rho_init=params["posterior_rho_init"], bias=d.bias is not None)
    if params["moped_enable"]:
        print("WARNING: MOPED method is not supported for LSTM layers!!!")
    bnn_layer.dnn_to_bnn_flag = True
    return bnn_layer


def dnn_to_bnn(m, bnn_prior_parameters):
    for name, child in list(m._modules.items()):
        if child is None:
            continue
        cname = child.__class__.__name__
        if child._modules:
            dnn_to_bnn(child, bnn_prior_parameters)
        elif "Conv" in cname:
            setattr(m, name, bnn_conv_layer(bnn_prior_parameters, child).to(child.weight.device))
        elif "Linear" in cname:
            setattr(m, name, bnn_linear_layer(bnn_prior_parameters, child).to(child.weight.device))
        elif "LSTM" in cname:
            setattr(m, name, bnn_lstm_layer(bnn_prior_parameters, child).to(next(child.parameters()).device))
    return


def get_kl_loss(m):
    """Sum of `layer.kl_loss()` over every module that has one (None for a model without Bayesian layers).  CUDA
    models: ONE launch for all tensors of all layers (btx_kl_gauss_model), differentiable (btx_kl_gauss_model_bwd)."""
    layers = [layer for layer in m.modules() if hasattr(layer, "kl_loss")]
    if not layers:
        return None
    if all(hasattr(layer, "_w") and layer._use_hip(layer._w()[0]) for layer in layers):
        from .. import autograd as _ag
        return _ag.kl_of_layers(layers)
    terms = [layer.kl_loss() for layer in layers]
    if len(terms) == 1:
        return terms[0]
    kl = terms[0]  # the reference's sequential += chain, bit for bit on CPU
    for t in terms[1:]:
        kl = kl + t
    return kl
