"""LSTMFlipout (reference layers/flipout_layers/rnn_flipout.py:46-153)."""
from ..base_variational_layer import _VariationalLSTM
from .linear_flipout import LinearFlipout

__all__ = ['LSTMFlipout']


class LSTMFlipout(_VariationalLSTM):
    """LSTM on two LinearFlipout layers — reference layers/flipout_layers/rnn_flipout.py:46-153."""
    _family = "flipout"
    _linear_cls = LinearFlipout
