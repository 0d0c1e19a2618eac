"""Conv{1,2,3}d / ConvTranspose{1,2,3}d Reparameterization (reference layers/variational_layers/conv_variational.py).

Thin public classes over `_VariationalNd` (../base_variational_layer.py).  Constructor signatures, attribute names,
parameter names and `forward(input, return_kl=True)` / `kl_loss()` follow the reference file cited per class.
"""
from ..base_variational_layer import _VariationalNd

__all__ = ['Conv1dReparameterization', 'Conv2dReparameterization', 'Conv3dReparameterization', 'ConvTranspose1dReparameterization', 'ConvTranspose2dReparameterization', 'ConvTranspose3dReparameterization']


class Conv1dReparameterization(_VariationalNd):
    """Conv1d with the reparameterization trick — reference layers/variational_layers/conv_variational.py:64-227."""
    _family, _nd, _transposed = "reparam", 1, False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init,
        self.posterior_rho_init = posterior_rho_init,
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, 0,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=True)


class Conv2dReparameterization(_VariationalNd):
    """Conv2d with the reparameterization trick — reference layers/variational_layers/conv_variational.py:230-402."""
    _family, _nd, _transposed = "reparam", 2, False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init,
        self.posterior_rho_init = posterior_rho_init,
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, 0,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=True)


class Conv3dReparameterization(_VariationalNd):
    """Conv3d with the reparameterization trick — reference layers/variational_layers/conv_variational.py:405-574."""
    _family, _nd, _transposed = "reparam", 3, False

    def __init__(self, in_channels, out_channels, kernel_size, prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init,
        self.posterior_rho_init = posterior_rho_init,
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, 0,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=True)


class ConvTranspose1dReparameterization(_VariationalNd):
    """ConvTranspose1d with the reparameterization trick — reference layers/variational_layers/conv_variational.py:577-744."""
    _family, _nd, _transposed = "reparam", 1, True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, output_padding=0, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init,
        self.posterior_rho_init = posterior_rho_init,
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, output_padding,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=True)


class ConvTranspose2dReparameterization(_VariationalNd):
    """ConvTranspose2d with the reparameterization trick — reference layers/variational_layers/conv_variational.py:747-919."""
    _family, _nd, _transposed = "reparam", 2, True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, output_padding=0, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init,
        self.posterior_rho_init = posterior_rho_init,
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, output_padding,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=True)


class ConvTranspose3dReparameterization(_VariationalNd):
    """ConvTranspose3d with the reparameterization trick — reference layers/variational_layers/conv_variational.py:922-1094."""
    _family, _nd, _transposed = "reparam", 3, True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, output_padding=0, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init,
        self.posterior_rho_init = posterior_rho_init,
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, output_padding,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=True)
