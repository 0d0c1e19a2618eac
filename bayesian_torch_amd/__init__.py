"""bayesian_torch_amd — the variational-layer forward hot path of IntelLabs/bayesian-torch, MI355X-native.

Drop-in surface (same names as the reference package `bayesian_torch`):
    bayesian_torch_amd.layers.{Linear,Conv1d,Conv2d,Conv3d,ConvTranspose1d,...}{Reparameterization,Flipout}
    bayesian_torch_amd.models.dnn_to_bnn.{dnn_to_bnn, get_kl_loss}
    bayesian_torch_amd.utils.util.get_rho
`install_alias()` registers this package under the name `bayesian_torch` so existing imports resolve here.

GPU tensors run through hand-written gfx950 kernels in libbtx.so (C-ABI: include/btx.h); see DESIGN.md.
"""
import sys

from . import _lib, functional, rng  # noqa: F401
from . import layers, models, utils  # noqa: F401
from .functional import set_precision, get_precision, set_output_layout  # noqa: F401
from .layers.base_variational_layer import set_backend  # noqa: F401
from .models.dnn_to_bnn import dnn_to_bnn, get_kl_loss  # noqa: F401
from .rng import manual_seed, set_sample_index, set_sample_lanes, assign_layer_ids, presample  # noqa: F401

__version__ = "0.1.0"


def install_alias(name="bayesian_torch"):
    """Make `import bayesian_torch...` resolve to this package (for code written against the reference)."""
    prefix = __name__
    for key, mod in list(sys.modules.items()):
        if key == prefix or key.startswith(prefix + "."):
            sys.modules[name + key[len(prefix):]] = mod
