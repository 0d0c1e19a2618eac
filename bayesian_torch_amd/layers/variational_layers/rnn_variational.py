"""LSTMReparameterization (reference layers/variational_layers/rnn_variational.py:46-153)."""
from ..base_variational_layer import _VariationalLSTM
from .linear_variational import LinearReparameterization

__all__ = ['LSTMReparameterization']


class LSTMReparameterization(_VariationalLSTM):
    """LSTM on two LinearReparameterization layers — reference layers/variational_layers/rnn_variational.py:46-153."""
    _family = "reparam"
    _linear_cls = LinearReparameterization
