"""Pure-torch functional restatement of the reference's variational forward + KL — TEST INFRASTRUCTURE ONLY.

Every function takes the noise EXPLICITLY (the reference draws it in place from torch's global generator) and
otherwise issues the same ATen calls in the same order as the reference method it cites, so on CPU it is bit-exact
against the imported reference when the noise is drawn with torch in the reference's order (pinned by
tests/golden/* which tools/make_golden.py generated from the reference itself).  bench.py times this module as the
``cpu_baseline`` ("port": same ATen kernels as the reference's CPU path; the reference tree is not on the GPU box).

Paths below are relative to /root/reference/bayesian_torch.
"""
import torch
import torch.nn.functional as F


def softplus(rho):
    """sigma = log1p(exp(rho)) — layers/variational_layers/linear_variational.py:145,160 (naive form, no threshold)."""
    return torch.log1p(torch.exp(rho))


def kl_div(mu_q, sigma_q, mu_p, sigma_p):
    """layers/base_variational_layer.py:53-68 — note the .mean()."""
    kl = torch.log(sigma_p) - torch.log(sigma_q) + (sigma_q ** 2 + (mu_q - mu_p) ** 2) / (2 * (sigma_p ** 2)) - 0.5
    return kl.mean()


def kl_loss(mu_w, rho_w, mu_b=None, rho_b=None, prior_mu=0.0, prior_sigma=1.0):
    """*.kl_loss() — e.g. layers/flipout_layers/conv_flipout.py:362-368; priors are full-shape filled buffers."""
    pm = torch.full_like(mu_w, prior_mu)
    ps = torch.full_like(mu_w, prior_sigma)
    kl = kl_div(mu_w, softplus(rho_w), pm, ps)
    if mu_b is not None:
        kl = kl + kl_div(mu_b, softplus(rho_b), torch.full_like(mu_b, prior_mu), torch.full_like(mu_b, prior_sigma))
    return kl


_CONV = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}
_CONVT = {1: F.conv_transpose1d, 2: F.conv_transpose2d, 3: F.conv_transpose3d}


def _contract(x, w, b, op):
    """op: dict(kind='linear') or dict(kind='conv'|'convT', nd, stride, padding, dilation, groups[, output_padding])."""
    if op["kind"] == "linear":
        return F.linear(x, w, b)
    if op["kind"] == "conv":
        return _CONV[op["nd"]](x, w, b, op["stride"], op["padding"], op["dilation"], op["groups"])
    return _CONVT[op["nd"]](x, w, b, op["stride"], op["padding"], op.get("output_padding", 0), op["groups"],
                            op["dilation"])


def reparam_forward(x, mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, op):
    """layers/variational_layers/conv_variational.py:361-380 (Conv2d; 1d :183-227, 3d :530-574, transpose :720-722)
    and layers/variational_layers/linear_variational.py:160-178."""
    weight = mu_w + (softplus(rho_w) * eps_w)
    bias = None
    if mu_b is not None:
        bias = mu_b + (softplus(rho_b) * eps_b)
    return _contract(x, weight, bias, op)


def flipout_forward(x, mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, sign_in, sign_out, op):
    """layers/flipout_layers/conv_flipout.py:376-417 and layers/flipout_layers/linear_flipout.py:149-174."""
    outputs = _contract(x, mu_w, mu_b, op)
    delta = softplus(rho_w) * eps_w
    bias = None
    if mu_b is not None:
        bias = softplus(rho_b) * eps_b
    perturbed = _contract(x * sign_in, delta, bias, op) * sign_out
    return outputs + perturbed


def draw_noise_like_reference(layer_family, x_shape, out_shape, w_shape, n_bias, generator=None):
    """Draw (eps_w, eps_b, sign_in, sign_out) with torch in the ORDER the reference consumes its global stream
    (SURVEY.md §0 fact 4):  reparam: eps_w, eps_b ; linear_flipout: eps_w, eps_b, s_in, s_out ;
    conv_flipout: s_in, s_out, eps_w, eps_b."""
    def nrm(shape):
        return torch.empty(shape).normal_(generator=generator)

    def sgn(shape):
        return torch.empty(shape).uniform_(-1, 1, generator=generator).sign()

    eps_b = None
    if layer_family == "reparam":
        eps_w = nrm(w_shape)
        if n_bias:
            eps_b = nrm(n_bias)
        return eps_w, eps_b, None, None
    if layer_family == "linear_flipout":
        eps_w = nrm(w_shape)
        if n_bias:
            eps_b = nrm(n_bias)
        return eps_w, eps_b, sgn(x_shape), sgn(out_shape)
    if layer_family == "conv_flipout":
        s_in, s_out = sgn(x_shape), sgn(out_shape)
        eps_w = nrm(w_shape)
        if n_bias:
            eps_b = nrm(n_bias)
        return eps_w, eps_b, s_in, s_out
    raise ValueError(layer_family)


def get_rho(sigma, delta):
    """utils/util.py:63-69 (MOPED)."""
    return torch.log(torch.expm1(delta * torch.abs(sigma)) + 1e-20)


# ---------------------------------------------------------------------------------------------------------------
# Whole-model CPU baseline: the reference's op chain for dnn_to_bnn(resnet) layers, on plain torch CPU kernels.
# ---------------------------------------------------------------------------------------------------------------
class RefVariationalConv2d(torch.nn.Module):
    """Module form of reparam_forward / flipout_forward that draws its noise the way the reference does
    (in place, global generator; conv_variational.py:361-364, conv_flipout.py:385-392).  CPU baseline only."""

    def __init__(self, mu, rho, stride, padding, kind, mu_b=None, rho_b=None):
        super().__init__()
        self.mu, self.rho, self.mu_b, self.rho_b = mu, rho, mu_b, rho_b
        self.stride, self.padding, self.kind = stride, padding, kind
        self.eps = torch.empty_like(mu)

    def forward(self, x):
        if self.kind == "Reparameterization":
            sigma = torch.log1p(torch.exp(self.rho))
            w = self.mu + sigma * self.eps.normal_()
            return F.conv2d(x, w, None, self.stride, self.padding)
        out = F.conv2d(x, self.mu, None, self.stride, self.padding)
        s_in = x.clone().uniform_(-1, 1).sign()
        s_out = out.clone().uniform_(-1, 1).sign()
        delta = torch.log1p(torch.exp(self.rho)) * self.eps.normal_()
        return out + F.conv2d(x * s_in, delta, None, self.stride, self.padding) * s_out


class RefVariationalLinear(torch.nn.Module):
    def __init__(self, mu, rho, mu_b, rho_b, kind):
        super().__init__()
        self.mu, self.rho, self.mu_b, self.rho_b, self.kind = mu, rho, mu_b, rho_b, kind
        self.eps, self.eps_b = torch.empty_like(mu), torch.empty_like(mu_b)

    def forward(self, x):
        sigma = torch.log1p(torch.exp(self.rho))
        sigma_b = torch.log1p(torch.exp(self.rho_b))
        if self.kind == "Reparameterization":
            return F.linear(x, self.mu + sigma * self.eps.normal_(), self.mu_b + sigma_b * self.eps_b.normal_())
        delta = sigma * self.eps.normal_()
        bias = sigma_b * self.eps_b.normal_()
        out = F.linear(x, self.mu, self.mu_b)
        s_in = x.clone().uniform_(-1, 1).sign()
        s_out = out.clone().uniform_(-1, 1).sign()
        return out + F.linear(x * s_in, delta, bias) * s_out


def convert_for_baseline(model):
    """Replace every variational layer of a CPU model (duck-typed: has mu_kernel/rho_kernel or mu_weight/rho_weight
    and a `_family`) by the Ref* module above, in place.  Used by bench.py's cpu_baseline leg only."""
    for name, child in list(model._modules.items()):
        if child is None:
            continue
        if hasattr(child, "mu_kernel") and hasattr(child, "rho_kernel"):
            kind = "Flipout" if getattr(child, "_family", "") == "flipout" else "Reparameterization"
            if child.mu_bias is not None or child.groups != 1 or child.dilation not in (1, (1, 1)):
                raise NotImplementedError("baseline converter covers the bias-free ResNet convs")
            # plain row-major copies: the product stores its parameters GEMM-major (a strided view), the reference hands
            # F.conv2d a contiguous [Cout, Cin, kh, kw] weight — same tensor layout => same MKLDNN path as the reference
            plain = lambda t: t.detach().clone(memory_format=torch.contiguous_format)  # noqa: E731
            setattr(model, name, RefVariationalConv2d(plain(child.mu_kernel), plain(child.rho_kernel), child.stride,
                                                      child.padding, kind))
        elif hasattr(child, "mu_weight") and hasattr(child, "rho_weight"):
            kind = "Flipout" if getattr(child, "_family", "") == "flipout" else "Reparameterization"
            setattr(model, name, RefVariationalLinear(child.mu_weight.detach().clone(), child.rho_weight.detach().clone(),
                                                      child.mu_bias.detach(), child.rho_bias.detach(), kind))
        else:
            convert_for_baseline(child)
    return model
