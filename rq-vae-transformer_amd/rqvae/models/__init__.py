"""rqvae/models/__init__.py:20-37 of the reference: ``create_model(config, ema=False)``."""

# Everything OUTSIDE the accelerated path (metrics, datasets, trainers, EMA wrapper, writers, ...) is not re-implemented:
# when a checkout of the reference is importable as well (its root later on sys.path, or RQVAE_REFERENCE_ROOT), those
# sub-modules resolve to the reference's own files, so the unchanged drivers keep working (`rqvae.metrics.fid`,
# `rqvae.img_datasets`, ...).  Modules that exist here always win: this directory stays first in __path__.
import os as _os
import pkgutil as _pkgutil

__path__ = _pkgutil.extend_path(__path__, __name__)
_ref = _os.environ.get('RQVAE_REFERENCE_ROOT')
if _ref:
    _cand = _os.path.join(_ref, *__name__.split('.'))
    if _os.path.isdir(_cand) and _cand not in __path__:
        __path__.append(_cand)

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
