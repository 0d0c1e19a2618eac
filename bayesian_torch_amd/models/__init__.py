from .dnn_to_bnn import dnn_to_bnn, get_kl_loss  # noqa: F401
