from .conv_flipout import *
from .linear_flipout import *
from .rnn_flipout import *
