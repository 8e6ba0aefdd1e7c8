"""rqvae/models/rqtransformer/__init__.py of the reference."""
from .transformers import RQTransformer


def get_rqtransformer(config):
    return RQTransformer(config)
