"""rqvae/models/rqvae/__init__.py:17-25 of the reference."""
from .rqvae import RQVAE


def get_rqvae(config):
    hps = config.hparams
    ddconfig = config.ddconfig
    return RQVAE(**hps, ddconfig=ddconfig, checkpointing=config.checkpointing)
