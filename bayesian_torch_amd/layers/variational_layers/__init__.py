from .conv_variational import *
from .linear_variational import *
