"""LinearReparameterization (reference layers/variational_layers/linear_variational.py:54-201).

Thin public classes over `_VariationalNd` (../base_variational_layer.py).  Constructor signatures, attribute names,
parameter names and `forward(input, return_kl=True)` / `kl_loss()` follow the reference file cited per class.
"""
from ..base_variational_layer import _VariationalNd

__all__ = ['LinearReparameterization']


class LinearReparameterization(_VariationalNd):
    """Linear with the reparameterization trick — reference layers/variational_layers/linear_variational.py:54-201."""
    _family, _nd, _transposed = "reparam", 0, False

    def __init__(self, in_features, out_features, prior_mean=0, prior_variance=1, posterior_mu_init=0,
                 posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.posterior_mu_init = posterior_mu_init,
        self.posterior_rho_init = posterior_rho_init,
        self._setup(in_features, out_features, 1, 1, 0, 1, 1, 0, prior_mean, prior_variance,
                    posterior_mu_init, posterior_rho_init, bias, check_groups=False)
