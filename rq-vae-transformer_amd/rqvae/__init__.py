"""MI355X-native drop-in for the sampling path of kakaobrain/rq-vae-transformer's ``rqvae`` package.

Same import surface as the reference for that path:
``rqvae.models.create_model``, ``rqvae.models.rqvae.RQVAE``, ``rqvae.models.rqtransformer.RQTransformer``,
``rqvae.utils.{config,dist,utils}``.  All arithmetic runs in librqamd.so (hand-written HIP for gfx950,
include/rqamd.h); these modules hold parameters under the reference's state_dict names and marshal
pointers.  There is no CPU fallback."""

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
