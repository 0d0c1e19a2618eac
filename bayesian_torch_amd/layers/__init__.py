"""bayesian_torch_amd.layers — same export surface as the reference `bayesian_torch/layers/__init__.py:1-6`
(the variational-forward hot path and the LSTM wrappers over its Linear layers: no quantized / tuple-passing wrappers — SURVEY.md §2 scope)."""
from . import variational_layers
from . import flipout_layers
from .variational_layers import *
from .flipout_layers import *
from .base_variational_layer import BaseVariationalLayer_, get_kernel_size, set_backend
