"""rqvae/models/__init__.py:20-37 of the reference: ``create_model(config, ema=False)``."""
from .rqvae import get_rqvae
from .rqtransformer import get_rqtransformer


def create_model(config, ema=False):
    model_type = config.type.lower()
    if model_type == 'rq-transformer':
        model = get_rqtransformer(config)
    elif model_type == 'rq-vae':
        model = get_rqvae(config)
    else:
        raise ValueError(f'{model_type} is invalid..')
    if ema:
        # reference: ExponentialMovingAverage wrapper (rqvae/models/ema.py) -- training only; every
        # sampling driver passes ema=False (main_sampling_fid.py:151).  Out of scope.
        raise NotImplementedError('EMA model wrappers are training-side and not part of the sampling path')
    return model, None
