"""Conv{1,2,3}d / ConvTranspose{1,2,3}d Flipout (reference layers/flipout_layers/conv_flipout.py).

Thin public classes over `_VariationalNd` (../base_variational_layer.py).  Constructor signatures, attribute names,
parameter names and `forward(input, return_kl=True)` / `kl_loss()` follow the reference file cited per class.
"""
from ..base_variational_layer import _VariationalNd

__all__ = ['Conv1dFlipout', 'Conv2dFlipout', 'Conv3dFlipout', 'ConvTranspose1dFlipout', 'ConvTranspose2dFlipout', 'ConvTranspose3dFlipout']


class Conv1dFlipout(_VariationalNd):
    """Conv1d with Flipout — reference layers/flipout_layers/conv_flipout.py:57-244."""
    _family, _nd, _transposed = "flipout", 1, False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init
        self.posterior_rho_init = posterior_rho_init
        self.kl = 0
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, 0,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=False)


class Conv2dFlipout(_VariationalNd):
    """Conv2d with Flipout — reference layers/flipout_layers/conv_flipout.py:247-439."""
    _family, _nd, _transposed = "flipout", 2, False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init
        self.posterior_rho_init = posterior_rho_init
        self.kl = 0
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, 0,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=False)


class Conv3dFlipout(_VariationalNd):
    """Conv3d with Flipout — reference layers/flipout_layers/conv_flipout.py:443-637."""
    _family, _nd, _transposed = "flipout", 3, False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init
        self.posterior_rho_init = posterior_rho_init
        self.kl = 0
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, 0,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=False)


class ConvTranspose1dFlipout(_VariationalNd):
    """ConvTranspose1d with Flipout — reference layers/flipout_layers/conv_flipout.py:640-831."""
    _family, _nd, _transposed = "flipout", 1, True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, output_padding=0, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init
        self.posterior_rho_init = posterior_rho_init
        self.kl = 0
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, output_padding,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=False)


class ConvTranspose2dFlipout(_VariationalNd):
    """ConvTranspose2d with Flipout — reference layers/flipout_layers/conv_flipout.py:834-1030."""
    _family, _nd, _transposed = "flipout", 2, True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, dilation=1, groups=1, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init
        self.posterior_rho_init = posterior_rho_init
        self.kl = 0
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, output_padding,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=False)


class ConvTranspose3dFlipout(_VariationalNd):
    """ConvTranspose3d with Flipout — reference layers/flipout_layers/conv_flipout.py:1033-1228."""
    _family, _nd, _transposed = "flipout", 3, True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, dilation=1, groups=1, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init
        self.posterior_rho_init = posterior_rho_init
        self.kl = 0
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, output_padding,
                    prior_mean, prior_variance, posterior_mu_init, posterior_rho_init, bias,
                    check_groups=False)
