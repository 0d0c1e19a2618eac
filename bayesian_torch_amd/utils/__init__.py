from .util import get_rho, entropy, predictive_entropy, mutual_information  # noqa: F401
